// fusion_literal.hpp -- TEST INFRASTRUCTURE (checker only; nothing under newmsm_amd/, include/ or bench.py's timed path may include it).
//
// A plain restatement of the glue code of Fusion::optimize, to compare include/msmhip_fusion.hpp against:
//   * MapBinaryCost / MapBinaryModel follow DummyCostFunction / DiscreteModelDummy (I/Fusion/Fusion.h:14-117): every term of the reduced
//     function in a std::map<int, std::vector<double>>, values appended behind whatever the key already holds, convertenergies reading the
//     first two values of every node, a fresh pairs array per initialise();
//   * literal_fusion_optimize follows the label loop (I/Fusion/Fusion.h:122-244): three OpenMP loops of per-clique evaluator calls into
//     per-step vectors, the PBF filled node by node / pair by pair / triplet by triplet, HOCR, convert, FastPD, acceptance.
// The third-party pieces (the PBF class of I/ELC/ELC.h, FPD::FastPD) are template parameters here as well: they are licence-restricted and
// FSL-bound, so the tests plug in small stand-ins of their own (tests/cpp/mini_pbf.hpp) on BOTH sides of the comparison.
// Parity unpinned, like the rest of oracle/: the reference cannot be built in this environment.
#ifndef MSM_ORACLE_FUSION_LITERAL_HPP
#define MSM_ORACLE_FUSION_LITERAL_HPP

#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <utility>
#include <vector>

namespace msm_oracle {

template <class CostBase>
class MapBinaryCost : public CostBase {
public:
    MapBinaryCost() { this->m_num_labels = 2; }
    void setUnaryCost(int node, double cost0, double cost1) {  // Fusion.h:24-28
        std::vector<double> &v = unary_[node];                 // insert-if-absent, then append
        v.push_back(cost0);
        v.push_back(cost1);
    }
    void setPairwiseCost(int ind, double E00, double E01, double E10, double E11) {  // Fusion.h:30-36
        std::vector<double> &v = pair_[ind];
        v.push_back(E00);
        v.push_back(E01);
        v.push_back(E10);
        v.push_back(E11);
    }
    double computePairwiseCost(int pair, int labelA, int labelB) override {  // Fusion.h:38-47
        const std::vector<double> &v = pair_[pair];
        if (labelA == 0 && labelB == 0) return v[0];
        if (labelA == 0 && labelB == 1) return v[1];
        if (labelA == 1 && labelB == 0) return v[2];
        return v[3];
    }
    void convertenergies(int numNodes, int numPairs, int numLabels) {  // Fusion.h:49-61
        this->m_num_nodes = numNodes;
        this->m_num_labels = numLabels;
        this->m_num_pairs = numPairs;
        delete[] this->unarycosts;
        this->unarycosts = new double[(size_t)numNodes * numLabels];
        for (int i = 0; i < numLabels; i++)
            for (int j = 0; j < numNodes; j++) {
                const std::vector<double> &v = unary_[j];
                if ((size_t)i >= v.size()) throw std::runtime_error("MapBinaryCost: a variable without a linear term (the reference reads an empty vector here)");
                this->unarycosts[(size_t)i * numNodes + j] = v[i];
            }
    }
    void reset() {  // Fusion.h:63-66
        unary_.clear();
        pair_.clear();
    }

private:
    std::map<int, std::vector<double>> unary_, pair_;
};

template <class ModelBase, class CostBase>
class MapBinaryModel : public ModelBase {
public:
    MapBinaryModel() : cost_(std::make_shared<MapBinaryCost<CostBase>>()) {
        this->m_num_pairs = 0;
        this->m_num_nodes = 0;
        this->m_num_labels = 2;
    }
    std::shared_ptr<CostBase> getCostFunction() override { return cost_; }
    void AddNode(int num) { this->m_num_nodes = num; }  // Fusion.h:83
    void AddUnaryTerm(int node, double E0, double E1) { cost_->setUnaryCost(node, E0, E1); }
    void AddPairwiseTerm(int node1, int node2, double E00, double E01, double E10, double E11) {  // Fusion.h:89-95
        std::vector<int> &v = ids_[this->m_num_pairs];
        v.push_back(node1);
        v.push_back(node2);
        cost_->setPairwiseCost(this->m_num_pairs, E00, E01, E10, E11);
        this->m_num_pairs++;
    }
    void initialise() {  // Fusion.h:97-107 (the reference leaks the previous array; the checker frees it)
        this->initLabeling();
        cost_->convertenergies(this->m_num_nodes, this->m_num_pairs, 2);
        delete[] this->pairs;
        this->pairs = new int[(size_t)this->m_num_pairs * 2 + 1];
        for (int i = 0; i < this->m_num_pairs; i++) {
            this->pairs[2 * i] = ids_[i][0];
            this->pairs[2 * i + 1] = ids_[i][1];
        }
    }
    void reset() {  // Fusion.h:109-115
        ids_.clear();
        this->m_num_pairs = 0;
        this->m_num_nodes = 0;
        this->m_num_labels = 2;
        cost_->reset();
    }

private:
    std::map<int, std::vector<int>> ids_;
    std::shared_ptr<MapBinaryCost<CostBase>> cost_;
};

struct LiteralTrace {
    std::vector<double> step_energy;
    std::vector<int> nodes_changed;
    long steps_skipped = 0;
};

// I/Fusion/Fusion.h:122-244
template <class PBF, class Solver, class BinaryModel, class Energy>
double literal_fusion_optimize(Energy &energy, int numthreads, LiteralTrace *trace) {
    struct U { double b[2]; };
    struct P { double b[4]; };
    struct T { double b[8]; };
    const int *pairs = energy.getPairs();
    const int *triplets = energy.getTriplets();
    const int num_nodes = energy.getNumNodes();
    int *labeling = energy.getLabeling();
    auto dummy = std::make_shared<BinaryModel>();
    energy.evaluateTotalCostSum();
    (void)numthreads;
    for (int sweep = 0; sweep < 2; ++sweep)
        for (int label = 0; label < energy.getNumLabels(); ++label) {
            double sumlabeldiff = 0.0;
            PBF pbf;
            std::vector<U> unary_data(num_nodes);
#pragma omp parallel for num_threads(numthreads)
            for (int node = 0; node < num_nodes; ++node) {
                unary_data[node].b[0] = energy.computeUnaryCost(node, labeling[node]);
                unary_data[node].b[1] = energy.computeUnaryCost(node, label);
#pragma omp critical
                sumlabeldiff += std::abs(label - labeling[node]);
            }
            if (!(sumlabeldiff > 0)) {
                if (trace) trace->steps_skipped++;
                continue;
            }
            for (int node = 0; node < num_nodes; ++node) pbf.AddUnaryTerm(node, unary_data[node].b[0], unary_data[node].b[1]);
            std::vector<P> pair_data(energy.getNumPairs());
#pragma omp parallel for num_threads(numthreads)
            for (int pair = 0; pair < energy.getNumPairs(); ++pair) {
                const int nodeA = pairs[pair * 2], nodeB = pairs[pair * 2 + 1];
                pair_data[pair].b[0] = energy.computePairwiseCost(pair, labeling[nodeA], labeling[nodeB]);
                pair_data[pair].b[1] = energy.computePairwiseCost(pair, labeling[nodeA], label);
                pair_data[pair].b[2] = energy.computePairwiseCost(pair, label, labeling[nodeB]);
                pair_data[pair].b[3] = energy.computePairwiseCost(pair, label, label);
            }
            for (int pair = 0; pair < energy.getNumPairs(); ++pair)
                pbf.AddPairwiseTerm(pairs[pair * 2], pairs[pair * 2 + 1], pair_data[pair].b[0], pair_data[pair].b[1], pair_data[pair].b[2], pair_data[pair].b[3]);
            std::vector<T> triplet_data(energy.getNumTriplets());
#pragma omp parallel for num_threads(numthreads)
            for (int triplet = 0; triplet < energy.getNumTriplets(); ++triplet) {
                const int lA = labeling[triplets[triplet * 3]], lB = labeling[triplets[triplet * 3 + 1]], lC = labeling[triplets[triplet * 3 + 2]];
                const int cand[2][3] = {{lA, lB, lC}, {label, label, label}};
                for (int k = 0; k < 8; ++k)  // 000 .. 111, the first node in the highest bit
                    triplet_data[triplet].b[k] = energy.computeTripletCost(triplet, cand[k >> 2 & 1][0], cand[k >> 1 & 1][1], cand[k & 1][2]);
            }
            for (int triplet = 0; triplet < energy.getNumTriplets(); ++triplet) {
                int node_ids[3] = {triplets[triplet * 3], triplets[triplet * 3 + 1], triplets[triplet * 3 + 2]};
                pbf.AddHigherTerm(3, node_ids, triplet_data[triplet].b);
            }
            dummy->reset();
            PBF qpbf;
            pbf.toQuadratic(qpbf, pbf.maxID() + 1);
            qpbf.convert(*dummy, qpbf.maxID() + 1);
            pbf.clear();
            qpbf.clear();
            dummy->initialise();
            int *Labels = dummy->getLabeling();
            Solver opt(dummy, 5);
            const double newEnergy = opt.run();
            opt.getLabeling(Labels);
            int nodesChanged = 0;
            for (int node = 0; node < energy.getNumNodes(); ++node)
                if (labeling[node] != label)
                    if (Labels[node] == 1) {
                        labeling[node] = label;
                        nodesChanged++;
                    }
            if (trace) {
                trace->step_energy.push_back(newEnergy);
                trace->nodes_changed.push_back(nodesChanged);
            }
        }
    return energy.evaluateTotalCostSum();
}

}  // namespace msm_oracle

#endif
