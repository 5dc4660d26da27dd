// similarity_device.hpp -- per-point feature-vector similarity used by the multivariate cost functions
// (MultivariateNonLinearSRegDiscreteCostFunction::computeUnaryCost, M/DiscreteCostFunction.cpp:444-458, and the HO
// variant :601-618): sim(source features of one vertex, interpolated target features, per-dimension weights).
#pragma once

#include "geom.hpp"

namespace msm {

// target feature d of the sample = barycentric_interpolation of the three vertex rows f0,f1,f2 (vertex-major)
__device__ __forceinline__ double interp_feature(const double *f0, const double *f1, const double *f2, int d, double wa, double wb, double wc) {
    return wa * f0[d] + wb * f1[d] + wc * f2[d];
}

// ---- sparsesimkernel::DICE / genDICE, M/similarities.cpp:201-253 -------------------------------------------
// Both threshold the two vectors at their idx-th smallest value, idx = floor(percentile * n) (:204), and count the
// elements at or above the thresholds.  The reference sorts copies to find the thresholds; counting ranks gives
// the same element without a sort: v is the idx-th smallest iff #(x < v) <= idx < #(x <= v).  Weights are ignored.
__device__ __forceinline__ int dice_index(double percentile, int n) { return (int)floor(percentile * n); }

__device__ __forceinline__ double dice_value(int sim, int size_a, int size_b, int common) {
    if (sim == 4) return 1.0 - ((2.0 * common) / (size_a + size_b));  // :225
    const double sb2 = (double)size_b * (double)size_b;               // pow(size_B, 2), exact
    return 1.0 - (2.0 * ((common / sb2) / ((size_a + size_b) / sb2)));  // :252
}

template <class F>
__device__ __forceinline__ double kth_smallest(int n, int idx, F val) {
    for (int j = 0; j < n; ++j) {
        const double v = val(j);
        int less = 0, leq = 0;
        for (int i = 0; i < n; ++i) {
            const double x = val(i);
            less += x < v;
            leq += x <= v;
        }
        if (less <= idx && idx < leq) return v;
    }
    return __longlong_as_double(0x7ff8000000000000ll);  // idx out of range (percentile >= 1) or NaN data: undefined in the reference
}

template <class FA, class FB>
__device__ __forceinline__ double dice_serial(int sim, int n, double percentile, FA A, FB B) {
    const int idx = dice_index(percentile, n);
    const double ta = kth_smallest(n, idx, A), tb = kth_smallest(n, idx, B);
    int size_a = n, size_b = n, common = 0;
    for (int i = 0; i < n; ++i) {
        int ov = 1;
        if (A(i) < ta) {
            --size_a;
            ov = 0;
        }
        if (B(i) < tb) {
            --size_b;
            ov = 0;
        }
        common += ov;
    }
    return dice_value(sim, size_a, size_b, common);
}

// ---- the same by one wavefront over patches staged in memory (the unary patch kernels, the gMSM pairwise kernel)
__device__ __forceinline__ int wave_count(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ double wave_kth_smallest(const double *X, int P, int idx, int lane) {
    double t = __longlong_as_double(0x7ff8000000000000ll);
    for (int j0 = 0; j0 < P; j0 += 64) {  // wavefront-uniform loop: a lane per candidate value, ranks by counting
        const int j = j0 + lane;
        const double v = j < P ? X[j] : 0.0;
        int less = 0, leq = 0;
        for (int i = 0; i < P; ++i) {
            const double x = X[i];
            less += x < v;
            leq += x <= v;
        }
        const unsigned long long bal = __ballot(j < P && less <= idx && idx < leq);
        if (bal) {
            const int src = __ffsll((long long)bal) - 1;
            t = __shfl(v, src, 64);
            break;
        }
    }
    return t;
}

// DICE / genDICE of two patches by one wavefront (M/similarities.cpp:201-253; see similarity_device.hpp)
__device__ __forceinline__ double patch_dice(const double *A, const double *B, int P, int lane, int simmeasure, double percentile) {
    const int idx = dice_index(percentile, P);
    const double ta = wave_kth_smallest(A, P, idx, lane), tb = wave_kth_smallest(B, P, idx, lane);
    int da = 0, db = 0, cm = 0;  // elements below the thresholds, and elements at or above both
    for (int i = lane; i < P; i += 64) {
        const bool la = A[i] < ta, lb = B[i] < tb;
        da += la;
        db += lb;
        cm += !la && !lb;
    }
    const int size_a = P - wave_count(da), size_b = P - wave_count(db), common = wave_count(cm);
    return dice_value(simmeasure, size_a, size_b, common);
}

// serial over the D dimensions in the reference's order; the target values are recomputed in the second pass
// instead of being stored (3 loads + 5 flops each)
// Nsrc > 0: sfeat / cfw are D x Nsrc (rows x Nsrc) row-major, as the ABI takes them; Nsrc == 0: vertex-major copies
// (Nsrc x D, Nsrc x rows), where the D values of a vertex are contiguous.
__device__ __forceinline__ double feature_vector_similarity(int sim, double percentile, const double *sfeat, const double *cfw, int cfw_rows, int Nsrc,
                                                            int sv, int D, const double *f0, const double *f1, const double *f2, double wa, double wb,
                                                            double wc) {
    auto W = [&](int d) { return (cfw && cfw_rows >= d + 1) ? (Nsrc ? cfw[(size_t)d * Nsrc + sv] : cfw[(size_t)sv * cfw_rows + d]) : 1.0; };
    auto A = [&](int d) { return Nsrc ? sfeat[(size_t)d * Nsrc + sv] : sfeat[(size_t)sv * D + d]; };
    if (sim == 4 || sim == 5) return dice_serial(sim, D, percentile, A, [&](int d) { return interp_feature(f0, f1, f2, d, wa, wb, wc); });
    if (sim == 2) {  // sparsesimkernel::corr, M/similarities.cpp:129-158
        double prod = 0.0, varA = 0.0, varB = 0.0, meanA = 0.0, meanB = 0.0, sum = 0.0;
        for (int d = 0; d < D; ++d) sum += W(d);
        for (int d = 0; d < D; ++d) {
            meanA += W(d) * A(d);
            meanB += W(d) * interp_feature(f0, f1, f2, d, wa, wb, wc);
        }
        if (sum > 0.0) {
            meanA /= sum;
            meanB /= sum;
        }
        for (int d = 0; d < D; ++d) {
            const double w = W(d), da = A(d) - meanA, db = interp_feature(f0, f1, f2, d, wa, wb, wc) - meanB;
            prod += w * da * db;
            varA += w * da * da;
            varB += w * db * db;
        }
        if (sum > 0.0) {
            prod /= sum;
            varA /= sum;
            varB /= sum;
        }
        const double r = (varA == 0.0 || varB == 0.0) ? 0.0 : prod / (sqrt(varA) * sqrt(varB));
        return 1 - (1 + r) * 0.5;  // get_sim_for_min, M/similarities.h:51-52
    }
    double prod = 0.0;  // sparsesimkernel::SSD, M/similarities.cpp:179-188
    for (int d = 0; d < D; ++d) {
        const double df = A(d) - interp_feature(f0, f1, f2, d, wa, wb, wc);
        prod += W(d) * df * df;
    }
    return sqrt(prod) / D;
}

// ---- the same feature-vector similarity (SSD / correlation, D <= 64) by the eight lanes of a group ---------------
// Lane j (0..7) owns the dimensions j, j + 8, ...: the group reads the three target rows and the moving row
// (vertex-major copies) as contiguous 64-byte pieces, and the sums over the dimensions are 8-lane DPP reductions.
// All 64 lanes of the wavefront must call this together; groups with go == false ride along (their result is unused).
constexpr int kMvLanes = 8;
constexpr int kMvKeep = 8;  // dimensions per lane: D <= 64

// x / s -- as x * (1 / s) when s is a power of two with a normal reciprocal: the product is then the correctly rounded quotient, bit for bit, and an
// FP64 division is some 35 instructions.  The weights of an unweighted similarity sum to the number of dimensions (32 in the HCP configurations).
__device__ __forceinline__ double div_exact(double x, double s) {
    const long long b = __double_as_longlong(s);
    if ((b & 0x000fffffffffffffll) == 0 && b >= 0x0010000000000000ll && b <= 0x7fd0000000000000ll) return x * __longlong_as_double(0x7fe0000000000000ll - b);
    return x / s;
}

// v of lane (lane ^ kXor), kXor < 16, by DPP moves (the shuffle of __shfl_xor goes through the LDS pipe: 36 per sample and pass of
// the kernels below, which they wait for): lane ^ 1 and lane ^ 2 are quad permutations, lane ^ 4 two row shifts by four with complementary bank masks
template <int kXor>
__device__ __forceinline__ int dpp_xor_i(int x) {
    static_assert(kXor == 1 || kXor == 2 || kXor == 4 || kXor == 8, "within a row of sixteen lanes");
    if (kXor == 1) return __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
    if (kXor == 2) return __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
    if (kXor == 8) return __builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false); // row_ror:8: lane <- lane +- 8 within its row
    const int t = __builtin_amdgcn_update_dpp(x, x, 0x104, 0xf, 0x5, false);         // row_shl:4 into lanes 0-3 and 8-11 of each row: lane <- lane + 4
    return __builtin_amdgcn_update_dpp(t, x, 0x114, 0xf, 0xa, false);                // row_shr:4 into lanes 4-7 and 12-15: lane <- lane - 4
}
template <int kXor>
__device__ __forceinline__ double dpp_xor(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)dpp_xor_i<kXor>((int)(unsigned)b), hi = (unsigned)dpp_xor_i<kXor>((int)(unsigned)((unsigned long long)b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// sum over the eight lanes of a group, in every lane; the order (lane ^ 4, then ^ 2, then ^ 1) is part of the result
__device__ __forceinline__ double mv_group_sum(double v) {
    v += dpp_xor<4>(v);
    v += dpp_xor<2>(v);
    v += dpp_xor<1>(v);
    return v;
}

__device__ __forceinline__ double feature_vector_similarity8(int sim, bool go, int j, int D, const double *sa /*moving row, D values*/,
                                                             const double *cw /*weight row or nullptr*/, int cfw_rows, const double *f0, const double *f1,
                                                             const double *f2, double wa, double wb, double wc) {
    double A[kMvKeep], B[kMvKeep], W[kMvKeep];
#pragma unroll
    for (int k = 0; k < kMvKeep; ++k) {
        const int d = j + kMvLanes * k;
        const bool in = go && d < D;
        A[k] = in ? sa[d] : 0.0;
        B[k] = in ? wa * f0[d] + wb * f1[d] + wc * f2[d] : 0.0;  // barycentric_interpolation per dimension
        W[k] = !in ? 0.0 : ((cw && cfw_rows >= d + 1) ? cw[d] : 1.0);
    }
    if (sim == 2) {  // sparsesimkernel::corr over the D dimensions, M/similarities.cpp:129-158
        double sum = 0, ma = 0, mb = 0;
#pragma unroll
        for (int k = 0; k < kMvKeep; ++k) {
            sum += W[k];
            ma += W[k] * A[k];
            mb += W[k] * B[k];
        }
        sum = mv_group_sum(sum);
        ma = mv_group_sum(ma);
        mb = mv_group_sum(mb);
        if (sum > 0.0) {
            ma = div_exact(ma, sum);
            mb = div_exact(mb, sum);
        }
        double pr = 0, va = 0, vb = 0;
#pragma unroll
        for (int k = 0; k < kMvKeep; ++k) {
            const double da = A[k] - ma, db = B[k] - mb;
            pr += W[k] * da * db;
            va += W[k] * da * da;
            vb += W[k] * db * db;
        }
        pr = mv_group_sum(pr);
        va = mv_group_sum(va);
        vb = mv_group_sum(vb);
        if (sum > 0.0) {
            pr = div_exact(pr, sum);
            va = div_exact(va, sum);
            vb = div_exact(vb, sum);
        }
        const double rr = (va == 0.0 || vb == 0.0) ? 0.0 : pr / (sqrt(va) * sqrt(vb));
        return 1 - (1 + rr) * 0.5;
    }
    double pr = 0;  // sparsesimkernel::SSD, :179-188
#pragma unroll
    for (int k = 0; k < kMvKeep; ++k) {
        const double df = A[k] - B[k];
        pr += W[k] * df * df;
    }
    pr = mv_group_sum(pr);
    return sqrt(pr) / D;
}

// The same with 16-byte loads: lane j owns the dimension pairs (16 k + 2 j, 16 k + 2 j + 1), so one load instruction of the
// group reads 128 contiguous bytes of a row.  D even (rows of vertex-major arrays are then 16-byte aligned), D <= 64.  In two halves:
// what the eight lanes sum (Moments) and the scalar arithmetic that ends the measure (similarity_from_moments: 4 divisions + 2 square roots for the
// correlation, one of each for SSD -- about 200 FP64 instructions that do not need eight lanes: a kernel with many samples per wavefront keeps the
// moments and finishes 64 samples with one pass of these instructions instead of eight; same operations on the same values, same result).
struct Moments {
    double pr, va, vb, sum;
};
__device__ __forceinline__ double similarity_from_moments(int sim, int D, Moments m) {
    if (sim == 2) {
        if (m.sum > 0.0) {
            m.pr = div_exact(m.pr, m.sum);
            m.va = div_exact(m.va, m.sum);
            m.vb = div_exact(m.vb, m.sum);
        }
        const double rr = (m.va == 0.0 || m.vb == 0.0) ? 0.0 : m.pr / (sqrt(m.va) * sqrt(m.vb));
        return 1 - (1 + rr) * 0.5;
    }
    return sqrt(m.pr) / D;
}
template <int kPairs = 4>  // dimension pairs per lane: D <= 16 * kPairs (2 for the 32 features of the HCP configurations: half the arithmetic of 4)
__device__ __forceinline__ Moments feature_vector_moments8x2(int sim, bool go, int j, int D, const double *sa, const double *cw, int cfw_rows, const double *f0,
                                                             const double *f1, const double *f2, double wa, double wb, double wc) {
    double A[2 * kPairs], B[2 * kPairs], W[2 * kPairs];
#pragma unroll
    for (int k = 0; k < kPairs; ++k) {
        const int d = 16 * k + 2 * j;
        const bool in = go && d < D;
        const double2 z = make_double2(0.0, 0.0);
        const double2 a2 = in ? *reinterpret_cast<const double2 *>(sa + d) : z;
        const double2 x0 = in ? *reinterpret_cast<const double2 *>(f0 + d) : z, x1 = in ? *reinterpret_cast<const double2 *>(f1 + d) : z,
                      x2 = in ? *reinterpret_cast<const double2 *>(f2 + d) : z;
        A[2 * k] = a2.x, A[2 * k + 1] = a2.y;
        B[2 * k] = in ? wa * x0.x + wb * x1.x + wc * x2.x : 0.0;  // barycentric_interpolation per dimension
        B[2 * k + 1] = in ? wa * x0.y + wb * x1.y + wc * x2.y : 0.0;
        W[2 * k] = !in ? 0.0 : ((cw && cfw_rows >= d + 1) ? cw[d] : 1.0);
        W[2 * k + 1] = !in ? 0.0 : ((cw && cfw_rows >= d + 2) ? cw[d + 1] : 1.0);
    }
    if (sim == 2) {  // sparsesimkernel::corr over the D dimensions, M/similarities.cpp:129-158
        double sum = 0, ma = 0, mb = 0;
#pragma unroll
        for (int k = 0; k < 2 * kPairs; ++k) {
            sum += W[k];
            ma += W[k] * A[k];
            mb += W[k] * B[k];
        }
        sum = mv_group_sum(sum);
        ma = mv_group_sum(ma);
        mb = mv_group_sum(mb);
        if (sum > 0.0) {
            ma = div_exact(ma, sum);
            mb = div_exact(mb, sum);
        }
        double pr = 0, va = 0, vb = 0;
#pragma unroll
        for (int k = 0; k < 2 * kPairs; ++k) {
            const double da = A[k] - ma, db = B[k] - mb;
            pr += W[k] * da * db;
            va += W[k] * da * da;
            vb += W[k] * db * db;
        }
        pr = mv_group_sum(pr);
        va = mv_group_sum(va);
        vb = mv_group_sum(vb);
        return Moments{pr, va, vb, sum};
    }
    double pr = 0;  // sparsesimkernel::SSD, :179-188
#pragma unroll
    for (int k = 0; k < 2 * kPairs; ++k) {
        const double df = A[k] - B[k];
        pr += W[k] * df * df;
    }
    pr = mv_group_sum(pr);
    return Moments{pr, 0.0, 0.0, 0.0};
}
__device__ __forceinline__ double feature_vector_similarity8x2(int sim, bool go, int j, int D, const double *sa, const double *cw, int cfw_rows, const double *f0,
                                                               const double *f1, const double *f2, double wa, double wb, double wc) {
    return similarity_from_moments(sim, D, feature_vector_moments8x2(sim, go, j, D, sa, cw, cfw_rows, f0, f1, f2, wa, wb, wc));
}

}  // namespace msm
