// msmhip_fusion.hpp -- the optimiser's side of the boundary in arrays.
//
// With the cost evaluations of a label step delivered as whole buffers (msm_cost_triplet_octets / msm_group_fusion_move), what is left of
// Fusion::optimize (I/Fusion/Fusion.h:122-244) on the host is (1) assembling the pseudo-Boolean function from 2 N + 4 P + 8 T numbers,
// (2) ELC / HOCR's reduction to a quadratic function, (3) copying that function into the binary model FastPD reads, (4) FastPD.  (2) and (4)
// are third-party code under research-only licences (I/ELC/ELC.h:5-8, I/FastPD/FastPD.h): they are template parameters here, used through
// the interface Fusion.h uses, and are neither reproduced nor modified.  (1) and (3) are the reference's own glue and this header replaces them:
//
//   * FlatBinaryCost / FlatBinaryModel: DummyCostFunction + DiscreteModelDummy (I/Fusion/Fusion.h:14-117) without std::map.  The
//     reference keeps every unary and pairwise term of the reduced function in a std::map<int, std::vector<double>> (one heap vector per
//     term); FastPD then reads the pairwise terms through computePairwiseCost(pair, l0, l1) -- a map look-up per PAIR() in its inner
//     loops (I/FastPD/FastPD.h:40,213-230,265-339), and convertenergies() walks the unary map label by label.  Here the terms live in
//     flat arrays that keep their capacity from one label step to the next; computePairwiseCost is an indexed load.  Observable
//     behaviour is the reference's: a node's FIRST AddUnaryTerm wins (PBF::convert ends with AddUnaryTerm(0, cnst, cnst), I/ELC/ELC.h:339,
//     which the map version appends behind node 0's two values and never reads), pairs keep the order of their AddPairwiseTerm calls,
//     unarycosts[label * numNodes + node] (FastPD aliases and modifies it, I/FastPD/FastPD.h:126,199), the labeling starts at 0.
//     One deliberate difference: a variable without a linear term gets (0, 0); the map version reads an empty vector there.
//   * fusion_optimize: the label loop of Fusion::optimize with the same calls in the same order on the PBF (AddUnaryTerm for every node,
//     AddPairwiseTerm for every pair, AddHigherTerm(3, ...) for every triplet, toQuadratic, convert, clear), the same solver calls
//     (Solver(model, 5), run(), getLabeling()) and the same acceptance rule -- so with the same PBF and solver types the labelings are the
//     same.  The costs come either through the per-clique evaluators (any DiscreteModel: the reference's three OpenMP loops) or, when the
//     energy offers whole steps (msmhip::FusionModel::labelStep, GroupFusionModel::labelStep), straight from the pinned buffers
//     the kernels wrote: no 2 N + 4 P + 8 T virtual calls, no per-step std::vector<TripletData>.
//
// Instantiation inside newMSM:
//     using BinaryCost  = msmhip::FlatBinaryCost<newmeshreg::DiscreteCostFunction>;
//     using BinaryModel = msmhip::FlatBinaryModel<newmeshreg::DiscreteModel, newmeshreg::DiscreteCostFunction>;
//     double e = msmhip::fusion_optimize<ELCReduce::PBF<double>, FPD::FastPD, BinaryModel>(model, verbose, numthreads);
// "Graph reuse across label steps" (the survey's other suggestion) is not possible from outside FastPD: its graph is built in its
// constructor from the model's pairs, and HOCR's auxiliary variables and edges depend on the signs of the cubic coefficients of the step.
#ifndef MSMHIP_FUSION_HPP
#define MSMHIP_FUSION_HPP

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <type_traits>
#include <utility>
#include <vector>

namespace msmhip {

// The binary cost function of a fusion move in arrays.  CostBase: newmeshreg::DiscreteCostFunction (M/DiscreteCostFunction.h:32-80) or
// any class with its protected members m_num_nodes / m_num_labels / m_num_pairs / unarycosts (delete[]d by the base) and virtual
// computePairwiseCost.
template <class CostBase>
class FlatBinaryCost : public CostBase {
public:
    FlatBinaryCost() { this->m_num_labels = 2; }

    void setNodeCount(int n) {  // AddNode: all nodes known before the terms arrive
        if ((size_t)n > seen_.size()) {
            seen_.resize((size_t)n, 0);
            unary_.resize(2 * (size_t)n, 0.0);
        }
        nodes_ = std::max(nodes_, n);
    }
    void setUnaryCost(int node, double cost0, double cost1) {  // Fusion.h:24-28: insert() keeps the first entry of a node
        if (node >= nodes_) setNodeCount(node + 1);
        if (seen_[(size_t)node]) return;
        seen_[(size_t)node] = 1;
        unary_[2 * (size_t)node] = cost0;
        unary_[2 * (size_t)node + 1] = cost1;
    }
    void setPairwiseCost(int ind, double E00, double E01, double E10, double E11) {  // Fusion.h:30-36
        if (4 * (size_t)ind + 4 > pair_.size()) pair_.resize(std::max(pair_.size() * 2, 4 * (size_t)ind + 4));
        double *e = &pair_[4 * (size_t)ind];
        e[0] = E00;
        e[1] = E01;
        e[2] = E10;
        e[3] = E11;
    }
    double computePairwiseCost(int pair, int labelA, int labelB) override {  // Fusion.h:38-47 (anything but 00 / 01 / 10 reads E11)
        const double *e = &pair_[4 * (size_t)pair];
        if (labelA == 0 && labelB == 0) return e[0];
        if (labelA == 0 && labelB == 1) return e[1];
        if (labelA == 1 && labelB == 0) return e[2];
        return e[3];
    }
    void convertenergies(int numNodes, int numPairs, int numLabels) {  // Fusion.h:49-61
        this->m_num_nodes = numNodes;
        this->m_num_labels = numLabels;
        this->m_num_pairs = numPairs;
        const size_t need = (size_t)numNodes * numLabels;
        if (need > table_cap_) {  // the base class delete[]s unarycosts: same allocator, kept while it is large enough
            delete[] this->unarycosts;
            this->unarycosts = new double[need ? need : 1];
            table_cap_ = need;
        }
        if ((size_t)numNodes > seen_.size()) setNodeCount(numNodes);
        for (int i = 0; i < numLabels && i < 2; ++i)
            for (int j = 0; j < numNodes; ++j) this->unarycosts[(size_t)i * numNodes + j] = unary_[2 * (size_t)j + i];
    }
    void reset() {  // Fusion.h:63-66; capacities stay
        std::fill(seen_.begin(), seen_.begin() + nodes_, (unsigned char)0);
        std::fill(unary_.begin(), unary_.begin() + 2 * (size_t)nodes_, 0.0);
        nodes_ = 0;
    }

private:
    std::vector<unsigned char> seen_;
    std::vector<double> unary_, pair_;  // [node][2]; [pair][00 01 10 11]
    int nodes_ = 0;
    size_t table_cap_ = 0;
};

// DiscreteModelDummy (Fusion.h:70-117).  ModelBase: newmeshreg::DiscreteModel (M/DiscreteModel.h:31-88) or any class with its protected
// members m_num_nodes / m_num_labels / m_num_pairs / labeling / pairs (both delete[]d by the base), initLabeling() and a virtual
// getCostFunction() returning std::shared_ptr<CostBase>.
template <class ModelBase, class CostBase>
class FlatBinaryModel : public ModelBase {
public:
    FlatBinaryModel() : costfct(std::make_shared<FlatBinaryCost<CostBase>>()) {
        this->m_num_pairs = 0;
        this->m_num_nodes = 0;
        this->m_num_labels = 2;
    }
    std::shared_ptr<CostBase> getCostFunction() override { return costfct; }

    // what PBF::convert calls (I/ELC/ELC.h:322-340)
    void AddNode(int num) {
        this->m_num_nodes = num;
        costfct->setNodeCount(num);
    }
    void AddUnaryTerm(int node, double E0, double E1) { costfct->setUnaryCost(node, E0, E1); }
    void AddPairwiseTerm(int node1, int node2, double E00, double E01, double E10, double E11) {
        const size_t k = (size_t)this->m_num_pairs;
        if (2 * k + 2 > ids_.size()) ids_.resize(std::max(ids_.size() * 2, (size_t)1024));
        ids_[2 * k] = node1;
        ids_[2 * k + 1] = node2;
        costfct->setPairwiseCost(this->m_num_pairs, E00, E01, E10, E11);
        this->m_num_pairs++;
    }
    void initialise() {  // Fusion.h:97-107 (which allocates a new pairs array per label step and never frees the previous one)
        if ((size_t)this->m_num_nodes > labeling_cap_) {
            this->initLabeling();
            labeling_cap_ = (size_t)this->m_num_nodes;
        } else if (this->labeling) {
            std::fill(this->labeling, this->labeling + this->m_num_nodes, 0);
        }
        costfct->convertenergies(this->m_num_nodes, this->m_num_pairs, 2);
        const size_t need = 2 * (size_t)this->m_num_pairs;
        if (need > pairs_cap_) {
            delete[] this->pairs;
            this->pairs = new int[need];
            pairs_cap_ = need;
        }
        std::copy(ids_.begin(), ids_.begin() + need, this->pairs);
    }
    void reset() {  // Fusion.h:109-115
        this->m_num_pairs = 0;
        this->m_num_nodes = 0;
        this->m_num_labels = 2;
        costfct->reset();
    }

protected:
    std::vector<int> ids_;
    size_t pairs_cap_ = 0, labeling_cap_ = 0;
    std::shared_ptr<FlatBinaryCost<CostBase>> costfct;
};

// The costs of one label step as the kernels deliver them (msmhip.hpp declares the same struct: FusionModel::labelStep returns it).
// unary_table: unarycosts[label * num_nodes + node] or null (all unary costs 0: the group model); pair_quads[4 p + k], k = 00 01 10 11 as
// pair_data[p].buffer[k] of Fusion.h:170-173, or null without pairs; triplet_octets[8 t + k], k = 000 .. 111 as triplet_data[t].buffer[k]
// of Fusion.h:188-195, or null without triplets.
#ifndef MSMHIP_STEP_COSTS_DEFINED
#define MSMHIP_STEP_COSTS_DEFINED
struct StepCosts {
    const double *unary_table = nullptr;
    const double *pair_quads = nullptr;
    const double *triplet_octets = nullptr;
};
#endif

namespace detail {
template <class E, class = void>
struct has_label_step : std::false_type {};
template <class E>
struct has_label_step<E, std::void_t<decltype(std::declval<E &>().labelStep(0))>> : std::true_type {};

template <class Energy>
int num_labels_of(Energy &e) { return e.getNumLabels(); }
}  // namespace detail

struct FusionTrace {  // optional: what a run did, for tests and timing
    std::vector<double> step_energy;  // the solver's energy of every label step taken
    std::vector<int> nodes_changed;
    long steps_skipped = 0;
};

// Fusion::optimize (I/Fusion/Fusion.h:122-244).  PBF: ELCReduce::PBF<double>; Solver: FPD::FastPD (constructed from a
// std::shared_ptr<BinaryModel> and the iteration cap 5, run(), getLabeling(int*)); BinaryModel: FlatBinaryModel<...>; Energy: the
// DiscreteModel of the registration (getNumNodes / Labels / Pairs / Triplets, getLabeling, getPairs, getTriplets, evaluators,
// evaluateTotalCostSum) -- with labelStep(label) when it can deliver whole steps.
template <class PBF, class Solver, class BinaryModel, class Energy>
double fusion_optimize(Energy &energy, bool verbose = false, int numthreads = 1, FusionTrace *trace = nullptr) {
    const int *pairs = energy.getPairs();
    const int *triplets = energy.getTriplets();
    const int NUM_SWEEPS = 2, MAX_FPD_ITERS = 5;
    const int num_nodes = energy.getNumNodes(), num_pairs = energy.getNumPairs(), num_triplets = energy.getNumTriplets();
    int *labeling = energy.getLabeling();
    auto binary = std::make_shared<BinaryModel>();
    double lastEnergy = energy.evaluateTotalCostSum();
    std::vector<double> unary_data, pair_data, triplet_data;  // per-clique path only; sized once
    (void)numthreads;

    for (int sweep = 0; sweep < NUM_SWEEPS; ++sweep) {
        for (int label = 0; label < detail::num_labels_of(energy); ++label) {
            long sumlabeldiff = 0;  // Fusion.h:153: the step is skipped only when every node already has this label
            for (int node = 0; node < num_nodes; ++node) sumlabeldiff += std::abs(label - labeling[node]);
            StepCosts step;
            if constexpr (detail::has_label_step<Energy>::value) {
                if (sumlabeldiff > 0) step = energy.labelStep(label);  // one ABI call: the kernels write all 4 P + 8 T costs
            } else {
                // the reference's first loop runs before the test: 2 N unary evaluations whether or not the step is taken
                unary_data.resize(2 * (size_t)num_nodes);
#pragma omp parallel for num_threads(numthreads)
                for (int node = 0; node < num_nodes; ++node) {
                    unary_data[2 * (size_t)node] = energy.computeUnaryCost(node, labeling[node]);
                    unary_data[2 * (size_t)node + 1] = energy.computeUnaryCost(node, label);
                }
            }
            if (sumlabeldiff <= 0) {
                if (trace) trace->steps_skipped++;
                continue;
            }
            PBF pbf;
            if constexpr (detail::has_label_step<Energy>::value) {
                for (int node = 0; node < num_nodes; ++node) {
                    const double E0 = step.unary_table ? step.unary_table[(size_t)labeling[node] * num_nodes + node] : 0.0;
                    const double E1 = step.unary_table ? step.unary_table[(size_t)label * num_nodes + node] : 0.0;
                    pbf.AddUnaryTerm(node, E0, E1);
                }
                for (int pair = 0; pair < num_pairs; ++pair) {
                    const double *b = step.pair_quads + 4 * (size_t)pair;
                    pbf.AddPairwiseTerm(pairs[pair * 2], pairs[pair * 2 + 1], b[0], b[1], b[2], b[3]);
                }
                for (int triplet = 0; triplet < num_triplets; ++triplet) {
                    int node_ids[3] = {triplets[triplet * 3], triplets[triplet * 3 + 1], triplets[triplet * 3 + 2]};
                    double b[8];  // AddHigherTerm takes a mutable array
                    std::copy(step.triplet_octets + 8 * (size_t)triplet, step.triplet_octets + 8 * (size_t)triplet + 8, b);
                    pbf.AddHigherTerm(3, node_ids, b);
                }
            } else {
                for (int node = 0; node < num_nodes; ++node) pbf.AddUnaryTerm(node, unary_data[2 * (size_t)node], unary_data[2 * (size_t)node + 1]);
                pair_data.resize(4 * (size_t)num_pairs);
#pragma omp parallel for num_threads(numthreads)
                for (int pair = 0; pair < num_pairs; ++pair) {
                    const int nodeA = pairs[pair * 2], nodeB = pairs[pair * 2 + 1];
                    double *b = &pair_data[4 * (size_t)pair];
                    b[0] = energy.computePairwiseCost(pair, labeling[nodeA], labeling[nodeB]);
                    b[1] = energy.computePairwiseCost(pair, labeling[nodeA], label);
                    b[2] = energy.computePairwiseCost(pair, label, labeling[nodeB]);
                    b[3] = energy.computePairwiseCost(pair, label, label);
                }
                for (int pair = 0; pair < num_pairs; ++pair) {
                    const double *b = &pair_data[4 * (size_t)pair];
                    pbf.AddPairwiseTerm(pairs[pair * 2], pairs[pair * 2 + 1], b[0], b[1], b[2], b[3]);
                }
                triplet_data.resize(8 * (size_t)num_triplets);
#pragma omp parallel for num_threads(numthreads)
                for (int triplet = 0; triplet < num_triplets; ++triplet) {
                    const int a = labeling[triplets[triplet * 3]], b = labeling[triplets[triplet * 3 + 1]], c = labeling[triplets[triplet * 3 + 2]];
                    double *e = &triplet_data[8 * (size_t)triplet];
                    e[0] = energy.computeTripletCost(triplet, a, b, c);
                    e[1] = energy.computeTripletCost(triplet, a, b, label);
                    e[2] = energy.computeTripletCost(triplet, a, label, c);
                    e[3] = energy.computeTripletCost(triplet, a, label, label);
                    e[4] = energy.computeTripletCost(triplet, label, b, c);
                    e[5] = energy.computeTripletCost(triplet, label, b, label);
                    e[6] = energy.computeTripletCost(triplet, label, label, c);
                    e[7] = energy.computeTripletCost(triplet, label, label, label);
                }
                for (int triplet = 0; triplet < num_triplets; ++triplet) {
                    int node_ids[3] = {triplets[triplet * 3], triplets[triplet * 3 + 1], triplets[triplet * 3 + 2]};
                    pbf.AddHigherTerm(3, node_ids, &triplet_data[8 * (size_t)triplet]);
                }
            }

            binary->reset();
            PBF qpbf;
            pbf.toQuadratic(qpbf, pbf.maxID() + 1);  // HOCR: cubic terms -> quadratic with auxiliary variables
            qpbf.convert(*binary, qpbf.maxID() + 1);
            pbf.clear();
            qpbf.clear();

            binary->initialise();
            int *Labels = binary->getLabeling();
            Solver opt(binary, MAX_FPD_ITERS);
            const double newEnergy = opt.run();
            opt.getLabeling(Labels);

            int nodesChanged = 0;
            for (int node = 0; node < num_nodes; ++node)
                if (labeling[node] != label && Labels[node] == 1) {
                    labeling[node] = label;
                    nodesChanged++;
                }
            if (trace) {
                trace->step_energy.push_back(newEnergy);
                trace->nodes_changed.push_back(nodesChanged);
            }
            if (verbose) {
                std::cout << "  LAB " << label << ":\t" << lastEnergy << " -> " << newEnergy << " / " << nodesChanged / (double)num_nodes * 100 << "% CHN" << std::endl;
                lastEnergy = newEnergy;
            }
        }
    }
    return energy.evaluateTotalCostSum();
}

}  // namespace msmhip

#endif  // MSMHIP_FUSION_HPP
