// fusion_elc.cpp -- include/msmhip_fusion.hpp driven with the REFERENCE'S OWN reduction: ELCReduce::PBF<double> of I/ELC/ELC.h, included from
// /root/reference by include path (it needs only std headers; nothing is copied, and this file is only built where the reference is present:
// tests/test_cpp_fusion.py skips otherwise).  FPD::FastPD includes FSL headers (I/FastPD/FastPD.h:35-36) and cannot be compiled here: the
// binary solve is the same deterministic stand-in on both sides (tests/cpp/mini_pbf.hpp: MiniSolver).
//   1. one label step's function -- AddUnaryTerm / AddPairwiseTerm / AddHigherTerm with ascending ids, toQuadratic, convert -- into the
//      array model (msmhip::FlatBinaryModel) and into the map-based restatement of DiscreteModelDummy (oracle/fusion_literal.hpp): same
//      nodes, same pairs in the same order, same unary table, same pairwise look-ups;
//   2. the whole label loop: msmhip::fusion_optimize (per-clique evaluators and whole-step buffers) against literal_fusion_optimize, all
//      three with the real PBF: identical labelings, step energies and change counts.
// Prints one JSON line.
#include <cstdio>
#include <string>

#include <ELC/ELC.h>

#include "../../oracle/fusion_literal.hpp"
#include "mini_pbf.hpp"
#include "synth_energy.hpp"

using PBF = ELCReduce::PBF<double>;
using FlatModel = msmhip::FlatBinaryModel<mini::MockModelBase, mini::MockCostBase>;
using MapModel = msm_oracle::MapBinaryModel<mini::MockModelBase, mini::MockCostBase>;

static int check_converted_models(int N, int L, int label, long &aux, long &edges) {
    SynthEnergy e(N, L);
    for (int i = 0; i < N; ++i) e.labeling[(size_t)i] = (i * 7 + 3) % L;
    PBF pbf;
    for (int i = 0; i < N; ++i) pbf.AddUnaryTerm(i, e.computeUnaryCost(i, e.labeling[(size_t)i]), e.computeUnaryCost(i, label));
    for (int p = 0; p < e.getNumPairs(); ++p) {
        const int a = e.pairs[2 * (size_t)p], b = e.pairs[2 * (size_t)p + 1], la = e.labeling[(size_t)a], lb = e.labeling[(size_t)b];
        pbf.AddPairwiseTerm(a, b, e.computePairwiseCost(p, la, lb), e.computePairwiseCost(p, la, label), e.computePairwiseCost(p, label, lb),
                            e.computePairwiseCost(p, label, label));
    }
    for (int t = 0; t < e.getNumTriplets(); ++t) {
        int ids[3] = {e.triplets[3 * (size_t)t], e.triplets[3 * (size_t)t + 1], e.triplets[3 * (size_t)t + 2]};
        const int c[2][3] = {{e.labeling[(size_t)ids[0]], e.labeling[(size_t)ids[1]], e.labeling[(size_t)ids[2]]}, {label, label, label}};
        double E[8];
        for (int k = 0; k < 8; ++k) E[k] = e.computeTripletCost(t, c[k >> 2 & 1][0], c[k >> 1 & 1][1], c[k & 1][2]);
        pbf.AddHigherTerm(3, ids, E);
    }
    PBF q;
    pbf.toQuadratic(q, pbf.maxID() + 1);
    FlatModel f;
    MapModel m;
    f.reset(), m.reset();
    q.convert(f, q.maxID() + 1);
    q.convert(m, q.maxID() + 1);
    f.initialise(), m.initialise();
    int bad = 0;
    bad += f.getNumNodes() != m.getNumNodes() || f.getNumPairs() != m.getNumPairs() || f.getNumLabels() != 2 || m.getNumLabels() != 2;
    bad += f.getNumNodes() <= N;  // HOCR added auxiliary variables
    aux = f.getNumNodes() - N, edges = f.getNumPairs();
    const int n = f.getNumNodes();
    for (int i = 0; i < 2 * n; ++i) bad += f.getCostFunction()->getUnaryCosts()[i] != m.getCostFunction()->getUnaryCosts()[i];
    for (int p = 0; p < f.getNumPairs(); ++p) {
        bad += f.getPairs()[2 * p] != m.getPairs()[2 * p] || f.getPairs()[2 * p + 1] != m.getPairs()[2 * p + 1];
        bad += !(f.getPairs()[2 * p] < f.getPairs()[2 * p + 1]);
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) bad += f.getCostFunction()->computePairwiseCost(p, a, b) != m.getCostFunction()->computePairwiseCost(p, a, b);
    }
    // the converted function IS the step's function: for a few assignments of the original variables, the minimum over nothing but the
    // model's own value with the auxiliary variables at their best (each appears in one cubic term's gadget: set greedily) is not checked
    // here -- that is ELC's business; what is checked is that both models hold the same numbers.
    return bad;
}

// What the reference's reduction costs per label step at the size of BASELINE configs 2 - 4 (ico4 control grid: 2 562 nodes, 5 120 control
// triangles, no pairs): AddUnaryTerm / AddHigherTerm for every clique, toQuadratic (HOCR), convert into the array model, initialise.  The costs
// are synthetic (hash), the sizes and the call sequence are those of I/Fusion/Fusion.h:157-217.  CPU only; the solve (FastPD) is not included.
#include <chrono>
static void time_reduction(int N, int T, int reps, double &assemble_ms, double &reduce_ms, double &convert_ms, long &aux, long &edges) {
    using clk = std::chrono::steady_clock;
    std::vector<int> trip((size_t)3 * T);
    for (int t = 0; t < T; ++t) {  // ascending triples spread over the nodes, each node in about six of them (an icosphere's valence)
        int a = (int)(((long long)t * 2654435761ll) % N), b = (a + 1 + t % 7) % N, c = (a + 9 + t % 11) % N;
        int v[3] = {a, b, c};
        std::sort(v, v + 3);
        if (v[0] == v[1]) v[1] = (v[1] + 1) % N;
        if (v[1] == v[2] || v[0] == v[2]) v[2] = (std::max(v[0], v[1]) + 1) % N;
        std::sort(v, v + 3);
        trip[3 * (size_t)t] = v[0], trip[3 * (size_t)t + 1] = v[1], trip[3 * (size_t)t + 2] = v[2];
    }
    FlatModel f;
    assemble_ms = reduce_ms = convert_ms = 0;
    for (int r = 0; r < reps; ++r) {
        auto t0 = clk::now();
        PBF pbf;
        for (int i = 0; i < N; ++i) pbf.AddUnaryTerm(i, hash_cost(1, i, r, 0, 0), hash_cost(1, i, r, 1, 0));
        for (int t = 0; t < T; ++t) {
            if (trip[3 * (size_t)t] == trip[3 * (size_t)t + 1] || trip[3 * (size_t)t + 1] == trip[3 * (size_t)t + 2]) continue;
            double E[8];
            for (int k = 0; k < 8; ++k) E[k] = 0.5 * hash_cost(3, t, r, k, 0);
            pbf.AddHigherTerm(3, &trip[3 * (size_t)t], E);
        }
        auto t1 = clk::now();
        PBF q;
        pbf.toQuadratic(q, pbf.maxID() + 1);
        auto t2 = clk::now();
        f.reset();
        q.convert(f, q.maxID() + 1);
        pbf.clear();
        q.clear();
        f.initialise();
        auto t3 = clk::now();
        assemble_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
        reduce_ms += std::chrono::duration<double, std::milli>(t2 - t1).count();
        convert_ms += std::chrono::duration<double, std::milli>(t3 - t2).count();
        aux = f.getNumNodes() - N, edges = f.getNumPairs();
    }
    assemble_ms /= reps, reduce_ms /= reps, convert_ms /= reps;
}

int main(int argc, char **argv) {
    const int threads = argc > 1 ? std::atoi(argv[1]) : 4;
    if (argc > 2 && std::string(argv[2]) == "time") {
        double a, r, c;
        long aux, edges;
        time_reduction(2562, 5120, 20, a, r, c, aux, edges);
        std::printf("{\"nodes\": 2562, \"triplets\": 5120, \"assemble_ms\": %.3f, \"toQuadratic_ms\": %.3f, \"convert_initialise_ms\": %.3f, \"aux_variables\": %ld, \"edges\": %ld}\n",
                    a, r, c, aux, edges);
        return 0;
    }
    long aux = 0, edges = 0;
    int bad_models = 0;
    for (int label = 0; label < 3; ++label) bad_models += check_converted_models(120, 7, label, aux, edges);

    const int N = 60, L = 7;
    SynthEnergy lit(N, L), per(N, L);
    SynthStepEnergy stp(N, L);
    msm_oracle::LiteralTrace tl;
    msmhip::FusionTrace tp, ts;
    const double el = msm_oracle::literal_fusion_optimize<PBF, mini::MiniSolver<MapModel>, MapModel>(lit, threads, &tl);
    const double ep = msmhip::fusion_optimize<PBF, mini::MiniSolver<FlatModel>, FlatModel>(per, false, threads, &tp);
    msmhip::fusion_optimize<PBF, mini::MiniSolver<FlatModel>, FlatModel>(stp, false, threads, &ts);
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
    std::printf("{\"bad_models\": %d, \"bad_drivers\": %d, \"steps\": %zu, \"skipped\": %ld, \"nodes_moved\": %d, \"energy_start\": %.17g, \"energy_end\": %.17g, "
                "\"aux_variables\": %ld, \"edges\": %ld}\n",
                bad_models, bad_drivers, tl.step_energy.size(), tl.steps_skipped, moved, zero.evaluateTotalCostSum(), el, aux, edges);
    return (bad_models || bad_drivers) ? 1 : 0;
}
