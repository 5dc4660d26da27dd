// host_mirror.cpp -- the C++ host side (include/msmhip.hpp) driving one discrete-optimisation iteration of a pairwise
// registration the way newMSM's model does (NonLinearSRegDiscreteModel::Initialize / setupCostFunction,
// M/DiscreteModel.cpp:63-108, :216-262), from inputs written by tests/test_cpp_host.py.  Writes the results back for
// comparison with the oracle.  No Python, no HIP headers: g++ + libmsmhip.so.
//
//   host_mirror <in.bin> <out.bin>
//
// File format (both ways): records "name dtype count\n" + raw little-endian payload, dtype f8 or i4.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>

#include "msmhip.hpp"

using namespace msmhip;

struct Bag {
    std::map<std::string, std::vector<double>> f;
    std::map<std::string, std::vector<int32_t>> i;
};

static Bag read_bag(const char *path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error(std::string("cannot open ") + path);
    Bag b;
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        std::istringstream hs(line);
        std::string name, dtype;
        size_t n;
        hs >> name >> dtype >> n;
        if (dtype == "f8") {
            auto &v = b.f[name];
            v.resize(n);
            in.read(reinterpret_cast<char *>(v.data()), (std::streamsize)(n * 8));
        } else {
            auto &v = b.i[name];
            v.resize(n);
            in.read(reinterpret_cast<char *>(v.data()), (std::streamsize)(n * 4));
        }
    }
    return b;
}

static void put(std::ofstream &out, const std::string &name, const std::vector<double> &v) {
    out << name << " f8 " << v.size() << "\n";
    out.write(reinterpret_cast<const char *>(v.data()), (std::streamsize)(v.size() * 8));
}
static void put(std::ofstream &out, const std::string &name, const std::vector<int32_t> &v) {
    out << name << " i4 " << v.size() << "\n";
    out.write(reinterpret_cast<const char *>(v.data()), (std::streamsize)(v.size() * 4));
}

int main(int argc, char **argv) {
    if (argc != 3) {
        std::fprintf(stderr, "usage: host_mirror in.bin out.bin\n");
        return 2;
    }
    try {
        Bag in = read_bag(argv[1]);
        std::ofstream out(argv[2], std::ios::binary);
        const int data_order = in.i["orders"][0], cp_order = in.i["orders"][1];

        // ---- the meshes of one resolution level; the regular spheres come from the library like in Mesh_registration
        auto [sphere, tri] = make_mesh_from_icosa(data_order);
        auto [control, ctri] = make_mesh_from_icosa(cp_order);
        Context ctx(0);
        Mesh TARGET(ctx, sphere, tri), SOURCE(ctx, sphere, tri), CPGRID(ctx, control, ctri);
        const int D = in.i["orders"][2];
        TARGET.set_pvalues(in.f["ref_feat"]);

        // ---- resampler free functions on the way in: project the control grid through the current warp
        // (Mesh_registration::project_CPgrid -> sphere_project_warp, M/mesh_registration.cpp:270-304)
        Points cp_now = sphere_project_warp(control, SOURCE, in.f["source_xyz"]);
        put(out, "cp_now", cp_now);

        // ---- NonLinearSRegDiscreteModel: cost function for this iteration
        Parameters P;
        P.kind = D > 1 ? MSM_COST_MULTIVARIATE : MSM_COST_UNIVARIATE;
        P.lambda = 0.2;
        DiscreteCostFunction costfct(ctx, P);
        costfct.set_meshes(TARGET, SOURCE, CPGRID);
        SOURCE.set_coords(in.f["source_xyz"]);
        CPGRID.set_coords(cp_now);
        costfct.reset_source(SOURCE);
        costfct.reset_CPgrid(CPGRID);
        costfct.set_featurespace(in.f["src_feat"], D);
        auto [MAXSEP, MVD] = cp_spacings(cp_now, ctri);
        costfct.set_spacings(MAXSEP, MVD);
        const Points &labels = in.f["labels"];
        const double centre[3] = {in.f["samples0"][0], in.f["samples0"][1], in.f["samples0"][2]};
        costfct.set_labels(labels, cp_rotations(centre, cp_now));
        const auto triplets = estimate_triplets(ctri);
        costfct.setTriplets(triplets);
        costfct.initialize((int)(control.size() / 3), (int)(labels.size() / 3), 0, (int)(triplets.size() / 3));
        costfct.get_source_data();
        put(out, "absw", costfct.AbsoluteWeights());

        // ---- what the optimisers ask for
        costfct.computeUnaryCosts();  // FastPD's height array
        put(out, "unarycosts", costfct.unarycosts);
        std::vector<double> one{costfct.computeUnaryCost(5, 3), costfct.computeTripletCost(7, 1, 2, 3)};
        put(out, "single", one);
        put(out, "triplet", costfct.computeTripletCost(in.i["tq_t"], in.i["tq_a"], in.i["tq_b"], in.i["tq_c"]));
        put(out, "octets", costfct.tripletOctets(in.i["labeling"], 4));
        put(out, "total", std::vector<double>{costfct.evaluateTotalCostSum(in.i["labeling"])});

        // ---- and on the way out: resample the moving data to the target sphere (metric_resample, R/resampler.cpp:304-309)
        put(out, "resampled", metric_resample(SOURCE, in.f["src_feat"], TARGET));

        // ---- after the optimiser: unfold the moved control grid (run_discrete_opt, M/mesh_registration.cpp:226-229)
        CPGRID.set_coords(in.f["folded_cp"]);
        int first_folded = 0;
        const int passes = unfold(CPGRID, 100.0, &first_folded);
        put(out, "unfold_counts", std::vector<int32_t>{passes, first_folded});
        put(out, "unfolded_cp", CPGRID.get_coords());
        Matrix normed = in.f["src_feat"];
        variance_normalise(normed, (int)(sphere.size() / 3));
        put(out, "normed", normed);

        // ---- error behaviour: the reference's exception text arrives in what()
        try {
            Parameters bad;
            bad.simmeasure = 3;
            DiscreteCostFunction nope(ctx, bad);
            return 1;
        } catch (const Error &e) {
            put(out, "error_code", std::vector<int32_t>{e.code});
            std::cout << "expected error: " << e.what() << "\n";
        }
        std::cout << "ok\n";
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "host_mirror failed: %s\n", e.what());
        return 1;
    }
}
