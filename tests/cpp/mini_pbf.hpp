// mini_pbf.hpp -- test stand-ins for the third-party pieces of a fusion move, so that the glue around them can be exercised here:
//   * MockCostBase / MockModelBase: classes with the members of newmeshreg::DiscreteCostFunction / DiscreteModel that the binary
//     models touch (M/DiscreteCostFunction.h:32-80, M/DiscreteModel.h:31-88);
//   * MiniPBF: a multilinear polynomial over binary variables with the interface Fusion.h uses of ELCReduce::PBF<double>
//     (AddUnaryTerm / AddPairwiseTerm / AddHigherTerm / maxID / toQuadratic / convert / clear).  Written from the published reduction
//     (H. Ishikawa, "Transformation of General Binary MRF Minimization to the First-Order Case", PAMI 2011, degree 3:
//       a < 0:  a xyz = min_w a w (x + y + z - 2);   a > 0:  a xyz = min_w a [ w (1 - x - y - z) + xy + yz + zx ] ),
//     not from I/ELC/ELC.h, whose ELC step, term order and numerics it does not try to match;
//   * MiniSolver: iterated conditional modes over the binary model through the accessors FPD::FastPD uses (getUnaryCosts(),
//     computePairwiseCost(pair, l0, l1), getPairs()), deterministic.  It is NOT FastPD: the tests compare two drivers that both use it.
#ifndef MSM_TESTS_MINI_PBF_HPP
#define MSM_TESTS_MINI_PBF_HPP

#include <algorithm>
#include <array>
#include <map>
#include <memory>
#include <vector>

namespace mini {

class MockCostBase {
public:
    virtual ~MockCostBase() { delete[] unarycosts; }
    double *getUnaryCosts() { return unarycosts; }
    virtual double computePairwiseCost(int, int, int) { return 0; }

protected:
    int m_num_nodes = 0, m_num_labels = 0, m_num_pairs = 0, m_num_triplets = 0;
    double *unarycosts = nullptr;
};

class MockModelBase {
public:
    virtual ~MockModelBase() {
        delete[] labeling;
        delete[] pairs;
    }
    int getNumNodes() const { return m_num_nodes; }
    int getNumLabels() const { return m_num_labels; }
    int getNumPairs() const { return m_num_pairs; }
    int *getLabeling() { return labeling; }
    const int *getPairs() const { return pairs; }
    virtual std::shared_ptr<MockCostBase> getCostFunction() = 0;

protected:
    void initLabeling() {
        if (m_num_nodes != 0) {
            delete[] labeling;
            labeling = new int[m_num_nodes];
            std::fill(labeling, labeling + m_num_nodes, 0);
        }
    }
    int m_num_nodes = 0, m_num_labels = 0, m_num_pairs = 0, m_num_triplets = 0;
    int *labeling = nullptr, *pairs = nullptr;
};

class MiniPBF {
public:
    using Key = std::array<int, 3>;  // ascending variable ids, unused slots = -1 in front
    void clear() {
        terms_.clear();
        constant_ = 0;
    }
    void AddUnaryTerm(int i, double E0, double E1) {
        constant_ += E0;
        add(E1 - E0, {i});
    }
    void AddPairwiseTerm(int i, int j, double E00, double E01, double E10, double E11) {
        constant_ += E00;
        add(E10 - E00, {i});
        add(E01 - E00, {j});
        add(E00 - E01 - E10 + E11, {i, j});
    }
    void AddHigherTerm(int n, int vars[], double E[]) {  // E[k]: bit (n - 1 - j) of k is the value of vars[j]
        for (int S = 0; S < (1 << n); ++S) {             // Moebius transform: the coefficient of the monomial over the set S
            double c = 0;
            for (int Tt = S;; Tt = (Tt - 1) & S) {
                c += ((__builtin_popcount(S) - __builtin_popcount(Tt)) & 1) ? -E[Tt] : E[Tt];
                if (Tt == 0) break;
            }
            std::vector<int> v;
            for (int j = 0; j < n; ++j)
                if (S >> (n - 1 - j) & 1) v.push_back(vars[j]);
            add(c, v);
        }
    }
    int maxID() const {
        int m = -1;
        for (const auto &t : terms_) m = std::max(m, t.first[2]);
        return m;
    }
    int toQuadratic(MiniPBF &q, int newvar) const {
        newvar = std::max(newvar, maxID() + 1);
        q.constant_ += constant_;
        for (const auto &t : terms_) {
            const Key &k = t.first;
            const double a = t.second;
            if (k[0] < 0) {
                q.terms_[k] += a;
                continue;
            }
            if (a == 0) continue;
            const int w = newvar++;
            if (a < 0) {
                for (int j = 0; j < 3; ++j) q.add(a, {k[j], w});
                q.add(-2 * a, {w});
            } else {
                for (int j = 0; j < 3; ++j) q.add(-a, {k[j], w});
                q.add(a, {w});
                q.add(a, {k[0], k[1]});
                q.add(a, {k[1], k[2]});
                q.add(a, {k[0], k[2]});
            }
        }
        return newvar;
    }
    template <class Optimizer>
    void convert(Optimizer &opt, int varcount) {  // the calls of PBF::convert, I/ELC/ELC.h:322-340
        varcount = std::max(varcount, maxID() + 1);
        opt.AddNode(varcount);
        for (const auto &t : terms_) {
            const Key &k = t.first;
            if (k[1] < 0) opt.AddUnaryTerm(k[2], 0, t.second);
            else opt.AddPairwiseTerm(k[1], k[2], 0, 0, 0, t.second);
        }
        opt.AddUnaryTerm(0, constant_, constant_);
    }
    double value(const std::vector<int> &x) const {
        double e = constant_;
        for (const auto &t : terms_) {
            bool on = true;
            for (int j = 0; j < 3; ++j)
                if (t.first[j] >= 0 && !x[(size_t)t.first[j]]) on = false;
            if (on) e += t.second;
        }
        return e;
    }
    size_t size() const { return terms_.size(); }

private:
    void add(double c, std::vector<int> v) {
        if (v.empty()) {
            constant_ += c;
            return;
        }
        std::sort(v.begin(), v.end());
        Key k = {-1, -1, -1};
        for (size_t j = 0; j < v.size(); ++j) k[3 - v.size() + j] = v[j];
        terms_[k] += c;
    }
    std::map<Key, double> terms_;
    double constant_ = 0;
};

// iterated conditional modes; Model: a binary model with the DiscreteModel seam
template <class Model>
class MiniSolver {
public:
    MiniSolver(std::shared_ptr<Model> m, int max_iters) : m_(std::move(m)), iters_(max_iters) {}
    double run() {
        const int n = m_->getNumNodes(), np = m_->getNumPairs();
        auto cost = m_->getCostFunction();
        const double *U = cost->getUnaryCosts();
        const int *pairs = m_->getPairs();
        std::vector<std::vector<int>> inc((size_t)n);
        for (int p = 0; p < np; ++p) {
            inc[(size_t)pairs[2 * p]].push_back(p);
            inc[(size_t)pairs[2 * p + 1]].push_back(p);
        }
        lab_.assign((size_t)n, 0);
        for (int it = 0; it < iters_; ++it) {
            bool changed = false;
            for (int v = 0; v < n; ++v) {
                double e[2];
                for (int l = 0; l < 2; ++l) {
                    e[l] = U[(size_t)l * n + v];
                    for (int p : inc[(size_t)v]) {
                        const int a = pairs[2 * p], b = pairs[2 * p + 1];
                        e[l] += cost->computePairwiseCost(p, a == v ? l : lab_[(size_t)a], b == v ? l : lab_[(size_t)b]);
                    }
                }
                const int best = e[1] < e[0] ? 1 : 0;
                if (best != lab_[(size_t)v]) {
                    lab_[(size_t)v] = best;
                    changed = true;
                }
            }
            if (!changed) break;
        }
        double total = 0;
        for (int v = 0; v < n; ++v) total += U[(size_t)lab_[(size_t)v] * n + v];
        for (int p = 0; p < np; ++p) total += cost->computePairwiseCost(p, lab_[(size_t)pairs[2 * p]], lab_[(size_t)pairs[2 * p + 1]]);
        return total;
    }
    void getLabeling(int *out) { std::copy(lab_.begin(), lab_.end(), out); }

private:
    std::shared_ptr<Model> m_;
    int iters_;
    std::vector<int> lab_;
};

}  // namespace mini

#endif
