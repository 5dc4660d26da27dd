// synth_energy.hpp -- a synthetic DiscreteModel-shaped multi-label energy (unary, pairwise and triplet costs from a hash) for the tests of
// include/msmhip_fusion.hpp: tests/cpp/fusion_flat.cpp (stand-in PBF) and tests/cpp/fusion_elc.cpp (the reference's own ELC.h).
#ifndef MSM_TESTS_SYNTH_ENERGY_HPP
#define MSM_TESTS_SYNTH_ENERGY_HPP

#include <algorithm>
#include <cstdint>
#include <vector>

#include "../../include/msmhip_fusion.hpp"

static double hash_cost(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t e) {
    uint64_t h = 0x9e3779b97f4a7c15ull;
    for (uint64_t v : {a, b, c, d, e}) {
        h ^= v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
        h *= 0xbf58476d1ce4e5b9ull;
        h ^= h >> 31;
    }
    return (double)(h >> 11) / (double)(1ull << 53) * 2.0 - 0.7;  // in (-0.7, 1.3): cubic coefficients of both signs
}

// a DiscreteModel-shaped energy: N nodes on a ring with chords, L labels
struct SynthEnergy {
    int N, L;
    std::vector<int> pairs, triplets, labeling;
    long calls = 0;
    SynthEnergy(int n, int l) : N(n), L(l), labeling((size_t)n, 0) {
        for (int i = 0; i < n; ++i) {
            pairs.push_back(std::min(i, (i + 1) % n));  // node ids ascending within a clique, as the reference's models list them
            pairs.push_back(std::max(i, (i + 1) % n));  // (M/DiscreteModel.cpp:286-289,303-306; ELC asserts i < j, I/ELC/ELC0.h:64)
            if (i % 3 == 0) {
                pairs.push_back(std::min(i, (i + 5) % n));
                pairs.push_back(std::max(i, (i + 5) % n));
            }
            int t[3] = {i, (i + 1) % n, (i + 2) % n};
            std::sort(t, t + 3);
            triplets.insert(triplets.end(), t, t + 3);
        }
    }
    int getNumNodes() const { return N; }
    int getNumLabels() const { return L; }
    int getNumPairs() const { return (int)pairs.size() / 2; }
    int getNumTriplets() const { return (int)triplets.size() / 3; }
    int *getLabeling() { return labeling.data(); }
    const int *getPairs() const { return pairs.data(); }
    const int *getTriplets() const { return triplets.data(); }
    double computeUnaryCost(int node, int label) { return hash_cost(1, node, label, 0, 0); }
    double computePairwiseCost(int pair, int a, int b) { return 0.3 * hash_cost(2, pair, a, b, 0); }
    double computeTripletCost(int t, int a, int b, int c) { return 0.5 * hash_cost(3, t, a, b, c); }
    double evaluateTotalCostSum() {
        double e = 0;
        for (int i = 0; i < N; ++i) e += computeUnaryCost(i, labeling[i]);
        for (int p = 0; p < getNumPairs(); ++p) e += computePairwiseCost(p, labeling[pairs[2 * p]], labeling[pairs[2 * p + 1]]);
        for (int t = 0; t < getNumTriplets(); ++t) e += computeTripletCost(t, labeling[triplets[3 * t]], labeling[triplets[3 * t + 1]], labeling[triplets[3 * t + 2]]);
        return e;
    }
};

// the same energy delivering whole label steps, as msmhip::FusionModel::labelStep does from the GPU
struct SynthStepEnergy : SynthEnergy {
    using SynthEnergy::SynthEnergy;
    std::vector<double> table, quads, octets;
    long steps = 0;
    msmhip::StepCosts labelStep(int label) {
        ++steps;
        if (table.empty()) {
            table.resize((size_t)L * N);
            for (int l = 0; l < L; ++l)
                for (int i = 0; i < N; ++i) table[(size_t)l * N + i] = SynthEnergy::computeUnaryCost(i, l);
        }
        quads.resize(4 * (size_t)getNumPairs());
        octets.resize(8 * (size_t)getNumTriplets());
        for (int p = 0; p < getNumPairs(); ++p) {
            const int c[2][2] = {{labeling[pairs[2 * p]], labeling[pairs[2 * p + 1]]}, {label, label}};
            for (int k = 0; k < 4; ++k) quads[4 * (size_t)p + k] = SynthEnergy::computePairwiseCost(p, c[k >> 1 & 1][0], c[k & 1][1]);
        }
        for (int t = 0; t < getNumTriplets(); ++t) {
            const int c[2][3] = {{labeling[triplets[3 * t]], labeling[triplets[3 * t + 1]], labeling[triplets[3 * t + 2]]}, {label, label, label}};
            for (int k = 0; k < 8; ++k) octets[8 * (size_t)t + k] = SynthEnergy::computeTripletCost(t, c[k >> 2 & 1][0], c[k >> 1 & 1][1], c[k & 1][2]);
        }
        msmhip::StepCosts s;
        s.unary_table = table.data();
        s.pair_quads = quads.data();
        s.triplet_octets = octets.data();
        return s;
    }
    // the per-clique evaluators must not be needed on this path
    double computeUnaryCost(int, int) { ++calls; return 0; }
    double computePairwiseCost(int, int, int) { ++calls; return 0; }
    double computeTripletCost(int, int, int, int) { ++calls; return 0; }
};

#endif
