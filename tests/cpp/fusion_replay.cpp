// fusion_replay.cpp -- the three OpenMP loops of Fusion::optimize (I/Fusion/Fusion.h:138-196), replayed verbatim against
// msmhip::FusionModel / GroupFusionModel: per label step 2 N computeUnaryCost, 4 P computePairwiseCost and 8 T computeTripletCost
// calls from 8 threads, exactly as the unmodified optimiser issues them.  The ELC reduction + FastPD solve between two steps
// (licence-restricted, FSL-bound) is replaced by a fixed pseudo-random acceptance of the proposed label, so that the labeling
// evolves the way it does under the optimiser.  Everything the loops collected goes to <out.bin> for comparison with the oracle
// (tests/test_cpp_host.py), together with the adapter's counters: one ABI call per label step, no clique evaluated on its own.
//
//   fusion_replay <in.bin> <out.bin>       file format: host_mirror.cpp
#include <omp.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>

#include "msmhip.hpp"
#include "msmhip_fusion.hpp"
#include "mini_pbf.hpp"

using namespace msmhip;

static std::map<std::string, std::vector<double>> F;
static std::map<std::string, std::vector<int32_t>> I;

static void read_bag(const char *path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error(std::string("cannot open ") + path);
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        std::istringstream hs(line);
        std::string name, dtype;
        size_t n;
        hs >> name >> dtype >> n;
        if (dtype == "f8") {
            F[name].resize(n);
            in.read(reinterpret_cast<char *>(F[name].data()), (std::streamsize)(n * 8));
        } else {
            I[name].resize(n);
            in.read(reinterpret_cast<char *>(I[name].data()), (std::streamsize)(n * 4));
        }
    }
}
template <class T>
static void put(std::ofstream &out, const std::string &name, const char *dtype, const std::vector<T> &v) {
    out << name << " " << dtype << " " << v.size() << "\n";
    out.write(reinterpret_cast<const char *>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}

// the buffers of I/Fusion/Fusion.h:11-13
struct UnaryData { double buffer[2]; };
struct PairData { double buffer[4]; };
struct TripletData { double buffer[8]; };

// Fusion::optimize's sweeps over `energy` (any class with DiscreteModel's interface), collecting what the loops read
template <class Model>
static void replay(Model &energy, int num_labels, int numthreads, int sweeps, std::vector<double> &unary_log, std::vector<double> &pair_log,
                   std::vector<double> &triplet_log, std::vector<int32_t> &labeling_log, std::vector<int32_t> &step_log) {
    const int *pairs = energy.getPairs();
    const int *triplets = energy.getTriplets();
    const int num_nodes = energy.getNumNodes();
    int *labeling = energy.getLabeling();
    unsigned rng = 12345u;
    for (int sweep = 0; sweep < sweeps; ++sweep) {
        for (int label = 0; label < num_labels; ++label) {
            double sumlabeldiff = 0.0;
            std::vector<UnaryData> unary_data(num_nodes);
#pragma omp parallel for num_threads(numthreads)
            for (int node = 0; node < num_nodes; ++node) {
                unary_data[node].buffer[0] = energy.computeUnaryCost(node, labeling[node]);
                unary_data[node].buffer[1] = energy.computeUnaryCost(node, label);
#pragma omp critical
                sumlabeldiff += std::abs(label - labeling[node]);
            }
            if (sumlabeldiff > 0) {
                std::vector<PairData> pair_data(energy.getNumPairs());
#pragma omp parallel for num_threads(numthreads)
                for (int pair = 0; pair < energy.getNumPairs(); ++pair) {
                    const int nodeA = pairs[pair * 2];
                    const int nodeB = pairs[pair * 2 + 1];
                    pair_data[pair].buffer[0] = energy.computePairwiseCost(pair, labeling[nodeA], labeling[nodeB]);
                    pair_data[pair].buffer[1] = energy.computePairwiseCost(pair, labeling[nodeA], label);
                    pair_data[pair].buffer[2] = energy.computePairwiseCost(pair, label, labeling[nodeB]);
                    pair_data[pair].buffer[3] = energy.computePairwiseCost(pair, label, label);
                }
                std::vector<TripletData> triplet_data(energy.getNumTriplets());
#pragma omp parallel for num_threads(numthreads)
                for (int triplet = 0; triplet < energy.getNumTriplets(); ++triplet) {
                    const int nodeA = triplets[triplet * 3];
                    const int nodeB = triplets[triplet * 3 + 1];
                    const int nodeC = triplets[triplet * 3 + 2];
                    triplet_data[triplet].buffer[0] = energy.computeTripletCost(triplet, labeling[nodeA], labeling[nodeB], labeling[nodeC]);  // 000
                    triplet_data[triplet].buffer[1] = energy.computeTripletCost(triplet, labeling[nodeA], labeling[nodeB], label);            // 001
                    triplet_data[triplet].buffer[2] = energy.computeTripletCost(triplet, labeling[nodeA], label, labeling[nodeC]);            // 010
                    triplet_data[triplet].buffer[3] = energy.computeTripletCost(triplet, labeling[nodeA], label, label);                      // 011
                    triplet_data[triplet].buffer[4] = energy.computeTripletCost(triplet, label, labeling[nodeB], labeling[nodeC]);            // 100
                    triplet_data[triplet].buffer[5] = energy.computeTripletCost(triplet, label, labeling[nodeB], label);                      // 101
                    triplet_data[triplet].buffer[6] = energy.computeTripletCost(triplet, label, label, labeling[nodeC]);                      // 110
                    triplet_data[triplet].buffer[7] = energy.computeTripletCost(triplet, label, label, label);                                // 111
                }
                step_log.push_back(label);
                labeling_log.insert(labeling_log.end(), labeling, labeling + num_nodes);
                for (const auto &u : unary_data) unary_log.insert(unary_log.end(), u.buffer, u.buffer + 2);
                for (const auto &p : pair_data) pair_log.insert(pair_log.end(), p.buffer, p.buffer + 4);
                for (const auto &t : triplet_data) triplet_log.insert(triplet_log.end(), t.buffer, t.buffer + 8);
                // in place of ELC + FastPD: a fixed fraction of the nodes takes the proposed label
                for (int node = 0; node < num_nodes; ++node) {
                    rng = rng * 1664525u + 1013904223u;
                    if (labeling[node] != label && (rng >> 24) % 3 == 0) labeling[node] = label;
                }
            }
        }
    }
}

// the same model with only the per-clique interface visible: msmhip::fusion_optimize then runs the reference's three loops against it
template <class Model>
struct PerClique {
    Model &m;
    int getNumNodes() const { return m.getNumNodes(); }
    int getNumLabels() const { return m.getNumLabels(); }
    int getNumPairs() const { return m.getNumPairs(); }
    int getNumTriplets() const { return m.getNumTriplets(); }
    int *getLabeling() { return m.getLabeling(); }
    const int *getPairs() const { return m.getPairs(); }
    const int *getTriplets() const { return m.getTriplets(); }
    double computeUnaryCost(int n, int l) { return m.computeUnaryCost(n, l); }
    double computePairwiseCost(int p, int a, int b) { return m.computePairwiseCost(p, a, b); }
    double computeTripletCost(int t, int a, int b, int c) { return m.computeTripletCost(t, a, b, c); }
    double evaluateTotalCostSum() { return m.evaluateTotalCostSum(); }
};

// A whole Fusion::optimize through msmhip::fusion_optimize (stand-in PBF and solver, tests/cpp/mini_pbf.hpp), twice from the zero
// labeling: from whole-step buffers (labelStep) and through the per-clique evaluators.  Same labelings, one ABI call per step taken.
template <class Model>
static void fused(Model &model, int threads, std::vector<int32_t> &labeling_out, std::vector<double> &info) {
    using BinaryModel = FlatBinaryModel<mini::MockModelBase, mini::MockCostBase>;
    const int n = model.getNumNodes();
    std::fill(model.getLabeling(), model.getLabeling() + n, 0);
    const long calls0 = model.counters.step_calls.load();
    FusionTrace ta, tb;
    const double ea = fusion_optimize<mini::MiniPBF, mini::MiniSolver<BinaryModel>, BinaryModel>(model, false, threads, &ta);
    const long calls_a = model.counters.step_calls.load() - calls0;
    std::vector<int32_t> la(model.getLabeling(), model.getLabeling() + n);
    std::fill(model.getLabeling(), model.getLabeling() + n, 0);
    PerClique<Model> pc{model};
    const long single0 = model.counters.single_calls.load();
    const double eb = fusion_optimize<mini::MiniPBF, mini::MiniSolver<BinaryModel>, BinaryModel>(pc, false, threads, &tb);
    std::vector<int32_t> lb(model.getLabeling(), model.getLabeling() + n);
    auto bits = [](const std::vector<double> &x, const std::vector<double> &y) {  // NaN costs (empty patch intersections) compare equal
        return x.size() == y.size() && (x.empty() || std::memcmp(x.data(), y.data(), x.size() * sizeof(double)) == 0);
    };
    const bool same = la == lb && bits(ta.step_energy, tb.step_energy) && ta.nodes_changed == tb.nodes_changed && bits({ea}, {eb});
    labeling_out = la;
    int moved = 0;
    for (int v : la) moved += v != 0;
    info = {same ? 1.0 : 0.0, (double)ta.step_energy.size(), (double)calls_a, ea, (double)moved, (double)(model.counters.single_calls.load() - single0)};
}

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    try {
        read_bag(argv[1]);
        const int mode = I["orders"][0], data_order = I["orders"][1], cp_order = I["orders"][2], D = I["orders"][3], threads = I["orders"][4],
                  sweeps = I["orders"][5];
        Context ctx(0);
        std::vector<double> unary_log, pair_log, triplet_log;
        std::vector<int32_t> labeling_log, step_log, counts;
        std::vector<double> totals, fused_info;
        std::vector<int32_t> fused_labeling;
        auto [dxyz, dtri] = make_mesh_from_icosa(data_order);
        auto [cxyz, ctri] = make_mesh_from_icosa(cp_order);
        if (mode == 0) {  // pairwise registration: one iteration's cost function as problem.build_cost assembles it
            Mesh TARGET(ctx, F["target_xyz"], dtri), SOURCE(ctx, dxyz, dtri), CPGRID(ctx, cxyz, ctri);
            TARGET.set_pvalues(F["ref_feat"]);
            Parameters P;
            P.kind = I["orders"][6];
            P.regularisermode = I["orders"][7];
            P.lambda = F["params"][0], P.shearmodulus = F["params"][1], P.bulkmodulus = F["params"][2], P.kexponent = F["params"][3], P.exponent = F["params"][4];
            DiscreteCostFunction costfct(ctx, P);
            costfct.set_meshes(TARGET, SOURCE, CPGRID);
            SOURCE.set_coords(F["source_xyz"]);
            CPGRID.set_coords(F["cp_xyz"]);
            costfct.reset_source(SOURCE);
            costfct.reset_CPgrid(CPGRID);
            costfct.set_featurespace(F["src_feat"], D);
            costfct.set_spacings(F["maxsep"], F["mvdmax"][0]);
            costfct.set_labels(F["labels"], F["rot"]);
            const std::vector<int32_t> triplets = I["triplets"], pairs = P.regularisermode == 1 ? I["pairs"] : std::vector<int32_t>();
            if (P.regularisermode == 1) costfct.setPairs(pairs);
            else costfct.setTriplets(triplets);
            costfct.get_source_data();
            FusionModel model(ctx, costfct, P.regularisermode == 1 ? std::vector<int32_t>() : triplets, pairs);
            model.setupCostFunction(P.regularisermode == 1);
            replay(model, model.getNumLabels(), threads, sweeps, unary_log, pair_log, triplet_log, labeling_log, step_log);
            totals.push_back(model.evaluateTotalCostSum());
            counts = {(int32_t)model.counters.step_calls.load(), (int32_t)model.counters.single_calls.load(), (int32_t)(model.counters.served.load() & 0x7fffffff)};
            fused(model, threads, fused_labeling, fused_info);
        } else {  // groupwise: S subjects as tests/test_gpu_group.py builds them
            const int S = I["orders"][6], L = (int)(F["labels"].size() / 3);
            GroupParameters GP;
            GP.lambda = F["params"][0];
            DiscreteGroupModel gm(ctx, GP, S);
            Mesh TEMPLATE(ctx, dxyz, dtri);
            gm.set_meshspace(TEMPLATE);
            gm.Initialize(cxyz, ctri);
            std::vector<std::unique_ptr<Mesh>> meshes;
            const size_t V = dxyz.size() / 3, N = cxyz.size() / 3;
            for (int s = 0; s < S; ++s) {
                meshes.emplace_back(new Mesh(ctx, dxyz, dtri));
                const Matrix feat(F["feat"].begin() + (size_t)s * D * V, F["feat"].begin() + (size_t)(s + 1) * D * V);
                gm.reset_meshspace(*meshes.back(), feat, D, s);
                meshes.back()->set_coords(Points(F["sph"].begin() + 3 * V * s, F["sph"].begin() + 3 * V * (s + 1)));
                gm.reset_meshspace(*meshes.back(), feat, D, s);
                gm.reset_CPgrid(Points(F["cp"].begin() + 3 * N * s, F["cp"].begin() + 3 * N * (s + 1)), s);
            }
            gm.set_labels(F["labels"]);
            GroupFusionModel model(ctx, gm);
            model.setupCostFunction();
            replay(model, L, threads, sweeps, unary_log, pair_log, triplet_log, labeling_log, step_log);
            counts = {(int32_t)model.counters.step_calls.load(), (int32_t)model.counters.single_calls.load(), (int32_t)(model.counters.served.load() & 0x7fffffff)};
            fused(model, threads, fused_labeling, fused_info);
        }
        std::ofstream out(argv[2], std::ios::binary);
        put(out, "total", "f8", totals);
        put(out, "unary", "f8", unary_log);
        put(out, "pairs", "f8", pair_log);
        put(out, "triplets", "f8", triplet_log);
        put(out, "labelings", "i4", labeling_log);
        put(out, "steps", "i4", step_log);
        put(out, "counts", "i4", counts);
        put(out, "fused_labeling", "i4", fused_labeling);
        put(out, "fused_info", "f8", fused_info);
        std::puts("ok");
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "fusion_replay failed: %s\n", e.what());
        return 1;
    }
}
