// fusion_flat.cpp -- include/msmhip_fusion.hpp (the fusion move's glue in arrays) against the map-based restatement of the reference's glue
// (oracle/fusion_literal.hpp), with the same stand-in PBF and solver on both sides (tests/cpp/mini_pbf.hpp).  No GPU, no libmsmhip:
//   1. a synthetic multi-label energy with unary, pairwise and triplet costs: the literal driver, msmhip::fusion_optimize through the
//      per-clique evaluators and msmhip::fusion_optimize through whole-step buffers must produce the same labelings, step energies and
//      numbers of changed nodes;
//   2. the binary models on their own: repeated / out-of-order AddUnaryTerm, the trailing AddUnaryTerm(0, c, c), capacity reuse;
//   3. the stand-in reduction is a reduction: min over the auxiliary variables of the quadratic function = the cubic function (brute force);
//   4. timing of what FastPD does with the model (convert -> initialise -> PAIR() look-ups) at a gMSM-like size, map against arrays.
// Prints one JSON line.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/msmhip_fusion.hpp"
#include "../../oracle/fusion_literal.hpp"
#include "mini_pbf.hpp"
#include "synth_energy.hpp"

using FlatModel = msmhip::FlatBinaryModel<mini::MockModelBase, mini::MockCostBase>;
using MapModel = msm_oracle::MapBinaryModel<mini::MockModelBase, mini::MockCostBase>;

static int check_models() {
    int bad = 0;
    FlatModel f;
    MapModel m;
    for (int round = 0; round < 3; ++round) {  // reused across rounds like across label steps, sizes going up and down
        const int n = round == 1 ? 9 : 5;
        f.reset();
        m.reset();
        auto fill = [&](auto &x) {
            x.AddNode(n);
            for (int i = n - 1; i >= 0; --i) x.AddUnaryTerm(i, 0, 1.5 * i + round);  // out of order
            x.AddUnaryTerm(2, 7, 9);                                                 // a second term for a node: ignored
            x.AddPairwiseTerm(0, 1, 0, 0, 0, -2.5);
            x.AddPairwiseTerm(3, 1, 1, 2, 3, 4 + round);
            x.AddUnaryTerm(0, 11, 11);  // convert()'s trailing constant: ignored as well, node 0 has its term
            x.initialise();
        };
        fill(f);
        fill(m);
        bad += f.getNumNodes() != m.getNumNodes() || f.getNumPairs() != m.getNumPairs() || f.getNumLabels() != 2;
        for (int i = 0; i < 2 * n; ++i) bad += f.getCostFunction()->getUnaryCosts()[i] != m.getCostFunction()->getUnaryCosts()[i];
        for (int p = 0; p < f.getNumPairs(); ++p) {
            bad += f.getPairs()[2 * p] != m.getPairs()[2 * p] || f.getPairs()[2 * p + 1] != m.getPairs()[2 * p + 1];
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) bad += f.getCostFunction()->computePairwiseCost(p, a, b) != m.getCostFunction()->computePairwiseCost(p, a, b);
        }
        for (int i = 0; i < n; ++i) bad += f.getLabeling()[i] != 0;
        f.getLabeling()[1] = 1;  // the next initialise() must clear it
    }
    // node 0 without a linear term: the trailing constant is its entry (both versions)
    f.reset();
    m.reset();
    f.AddNode(2), m.AddNode(2);
    f.AddUnaryTerm(1, 0, 3), m.AddUnaryTerm(1, 0, 3);
    f.AddUnaryTerm(0, 4, 4), m.AddUnaryTerm(0, 4, 4);
    f.initialise(), m.initialise();
    for (int i = 0; i < 4; ++i) bad += f.getCostFunction()->getUnaryCosts()[i] != m.getCostFunction()->getUnaryCosts()[i];
    return bad;
}

static int check_reduction() {  // min over the auxiliary variables of the quadratic function == the original, for every assignment
    int bad = 0;
    mini::MiniPBF pbf, q;
    int v0[3] = {0, 1, 2}, v1[3] = {1, 2, 3};
    double e0[8], e1[8];
    for (int k = 0; k < 8; ++k) e0[k] = hash_cost(9, k, 0, 0, 0), e1[k] = hash_cost(9, k, 1, 0, 0);
    pbf.AddHigherTerm(3, v0, e0);
    pbf.AddHigherTerm(3, v1, e1);
    pbf.AddUnaryTerm(3, 0.25, -0.5);
    pbf.AddPairwiseTerm(0, 3, 0.1, 0.2, 0.3, -0.4);
    const int nv = pbf.toQuadratic(q, pbf.maxID() + 1);
    for (int x = 0; x < 16; ++x) {
        std::vector<int> a((size_t)nv, 0);
        for (int j = 0; j < 4; ++j) a[(size_t)j] = x >> j & 1;
        // the original: sum of the tables
        double want = e0[a[0] * 4 + a[1] * 2 + a[2]] + e1[a[1] * 4 + a[2] * 2 + a[3]] + (a[3] ? -0.5 : 0.25);
        want += a[0] ? (a[3] ? -0.4 : 0.3) : (a[3] ? 0.2 : 0.1);
        if (std::abs(pbf.value(a) - want) > 1e-12) ++bad;
        double best = 1e300;
        for (int w = 0; w < (1 << (nv - 4)); ++w) {
            for (int j = 4; j < nv; ++j) a[(size_t)j] = w >> (j - 4) & 1;
            best = std::min(best, q.value(a));
        }
        if (std::abs(best - want) > 1e-12) ++bad;
    }
    return bad;
}

template <class Model>
static void time_model(int nodes, int npairs, int lookups, double &assemble_ms, double &lookup_ns, double &checksum) {
    using clk = std::chrono::steady_clock;
    Model m;
    uint64_t s = 88172645463325252ull;
    auto next = [&]() {
        s ^= s << 13, s ^= s >> 7, s ^= s << 17;
        return s;
    };
    assemble_ms = 1e300;
    for (int rep = 0; rep < 3; ++rep) {  // three label steps: the second and third reuse what the first allocated (arrays) or rebuild it (maps)
        auto t0 = clk::now();
        m.reset();
        m.AddNode(nodes);
        for (int i = 0; i < nodes; ++i) m.AddUnaryTerm(i, 0, 0.001 * i);
        for (int p = 0; p < npairs; ++p) m.AddPairwiseTerm((int)(next() % nodes), (int)(next() % nodes), 0, 0, 0, 0.5 + p % 7);
        m.AddUnaryTerm(0, 1, 1);
        m.initialise();
        assemble_ms = std::min(assemble_ms, std::chrono::duration<double, std::milli>(clk::now() - t0).count());
    }
    auto cost = m.getCostFunction();
    checksum = 0;
    auto t0 = clk::now();
    for (int i = 0; i < lookups; ++i) {
        const uint64_t r = next();
        checksum += cost->computePairwiseCost((int)(r % npairs), (int)(r >> 40 & 1), (int)(r >> 41 & 1));
    }
    lookup_ns = std::chrono::duration<double, std::nano>(clk::now() - t0).count() / lookups;
}

int main(int argc, char **argv) {
    const int threads = argc > 1 ? std::atoi(argv[1]) : 4;
    int bad_models = check_models(), bad_reduction = check_reduction();

    // 1. the three drivers on the same energy
    const int N = 60, L = 7;
    SynthEnergy lit(N, L), per(N, L);
    SynthStepEnergy stp(N, L);
    msm_oracle::LiteralTrace tl;
    msmhip::FusionTrace tp, ts;
    const double el = msm_oracle::literal_fusion_optimize<mini::MiniPBF, mini::MiniSolver<MapModel>, MapModel>(lit, threads, &tl);
    const double ep = msmhip::fusion_optimize<mini::MiniPBF, mini::MiniSolver<FlatModel>, FlatModel>(per, false, threads, &tp);
    msmhip::fusion_optimize<mini::MiniPBF, mini::MiniSolver<FlatModel>, FlatModel>(stp, false, threads, &ts);
    const double es = static_cast<SynthEnergy &>(stp).evaluateTotalCostSum();
    int bad_drivers = 0;
    bad_drivers += lit.labeling != per.labeling || lit.labeling != stp.labeling;
    bad_drivers += tl.step_energy != tp.step_energy || tl.step_energy != ts.step_energy;
    bad_drivers += tl.nodes_changed != tp.nodes_changed || tl.nodes_changed != ts.nodes_changed;
    bad_drivers += tl.steps_skipped != tp.steps_skipped || tl.steps_skipped != ts.steps_skipped;
    bad_drivers += el != ep || el != es;
    bad_drivers += stp.calls != 0 || stp.steps != (long)ts.step_energy.size();
    int moved = 0;
    for (int v : lit.labeling) moved += v != 0;
    SynthEnergy zero(N, L);
    const double e_start = zero.evaluateTotalCostSum();

    // 4. timing at the size of one label step of a 16-subject group at ico6 / ico4 after HOCR: 41 k nodes + 82 k auxiliary variables, 0.9 M edges
    double a_map, l_map, c_map, a_flat, l_flat, c_flat;
    const int tn = argc > 2 ? std::atoi(argv[2]) : 123000, tp_ = argc > 3 ? std::atoi(argv[3]) : 900000;
    time_model<MapModel>(tn, tp_, 2000000, a_map, l_map, c_map);
    time_model<FlatModel>(tn, tp_, 2000000, a_flat, l_flat, c_flat);

    std::printf("{\"bad_models\": %d, \"bad_reduction\": %d, \"bad_drivers\": %d, \"steps\": %zu, \"skipped\": %ld, \"nodes_moved\": %d, \"energy_start\": %.17g, "
                "\"energy_end\": %.17g, \"timing\": {\"nodes\": %d, \"pairs\": %d, \"assemble_ms_map\": %.3f, \"assemble_ms_flat\": %.3f, \"lookup_ns_map\": %.2f, "
                "\"lookup_ns_flat\": %.2f, \"checksums_equal\": %s}}\n",
                bad_models, bad_reduction, bad_drivers, tl.step_energy.size(), tl.steps_skipped, moved, e_start, el, tn, tp_, a_map, a_flat, l_map, l_flat,
                c_map == c_flat ? "true" : "false");
    return (bad_models || bad_reduction || bad_drivers) ? 1 : 0;
}
