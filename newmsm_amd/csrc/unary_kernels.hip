// unary_kernels.hip -- the unary label-cost table (computeUnaryCosts, M/DiscreteCostFunction.cpp:236-243)
// as three gfx950 kernels.
//
//   k_unary_samples   one workgroup per control point.  The control
//                     point's patch (source coordinates: the "neighbour ring") and its L rotation matrices are
//                     staged in LDS once and reused by all L*P point samples.  Each sample = rotate, find
//                     the nearest target triangle, interpolate (get_target_data, :353-376).  The search
//                     is split into three LDS-connected phases so that every phase keeps all 64 lanes of a
//                     wavefront busy with the same kind of work:
//                       A  per sample: dense-grid cell -> octree leaf, float cone filter over the leaf's
//                          entries; surviving (sample, triangle) pairs go to an LDS queue;
//                       B  per pair (evenly spread over the lanes): the exact FP64 inside test of the
//                          reference (project_point + point_in_triangle);
//                       C  per sample: exactly one triangle contains the projection (the normal case) ->
//                          barycentric interpolation of the reference feature, value written to HBM.
//                     Samples with zero or several containing triangles need the reference's tie-breaks
//                     and fallbacks; they are rare and are appended to a fix-up list instead of being
//                     handled here (keeps this kernel's register budget small = more waves per SIMD).
//   k_unary_fixup     re-does the listed samples with the complete search (search_device.hpp).
//   k_unary_reduce_*  one workgroup per control point: similarity of the moving patch with the L sampled
//                     target patches (wavefront shuffle reductions) -> U[label*N + node].
//
// Decisions (which triangle) use the reference's FP64 arithmetic; see search_device.hpp.
#include <algorithm>

#include "kernels.hpp"
#include "search_device.hpp"
#include "similarity_device.hpp"

#ifndef MSM_VARIANT
#define MSM_VARIANT 0  // profiling only: 1 = stop after phase A, 2 = stop after phase B, 4 = skip the cone filter
#endif

namespace msm {

namespace {

constexpr int kChunk = 768;       // samples per LDS pass (3 rounds of 256 lanes)
constexpr int kQueueCap = 3072;   // (sample, triangle) pairs per pass
constexpr int kDeferred = 1 << 20; // nin marker: leave this sample to the fix-up kernel
constexpr int kDone = -2;          // nin marker: already emitted by the fast path
constexpr unsigned kInvalidPair = 0xffffffffu;  // queue slot reserved by a sample that was deferred (sample ids stay below 1023)
constexpr int kTriBits = 22;      // queue entry = sample << 22 | triangle

__device__ __forceinline__ void raise_status(int *status, int code) { atomicMin(status, code); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

struct SamplesArgs {
    DevTree tree;
    const double *tfeat;   // target feature (D == 1: V doubles) or nullptr when only weights are wanted
    int N, L;
    const double *rnl;     // N x L x 9
    const double *src;     // 3 x Nsrc SoA
    int Nsrc;
    const int *pptr, *pidx;
    int pmax;
    int nsplit;            // workgroups per control point (label ranges)
    // outputs, indexed by global sample id g = L*pptr[node] + l*P + i
    double *tval;          // D == 1: interpolated reference feature
    int *stri;             // otherwise: triangle id and the three raw barycentric weights
    double *sw3;           // 3 doubles per sample
    unsigned long long *fix_list;  // node << 32 | local sample
    unsigned int *fix_count;
    unsigned int fix_cap;
    // fused reduction (univariate): moving feature, weights, AbsoluteWeights, output table
    const double *sfeat;   // Nsrc (feature row 1)
    const double *cfw;     // Nsrc (weight row 1) or nullptr
    const double *absw;    // N
    int simmeasure;
    double *U;             // L x N
    int *redo_list;        // nodes whose reduction must wait for the fix-up kernel
    unsigned int *redo_count;
    int *status;
    unsigned long long *nsamples;
};

// locate the octree leaf of p (same decisions as find_closest_triangle) and the sub-cell of p inside the leaf's
// box: the box is cut 4x4x4 and the three cut positions per axis are exact dyadics, so plain comparisons place p
// in the same (closed) sub-cell the mask builder reasoned about.
__device__ __forceinline__ int4 locate_leaf(const DevTree &T, const V3 &p, int &subcell) {
    const int G = 1 << T.grid_depth;
    const double h = 2 * kBounds / G;
    const int ix = grid_axis(p.x, G, h), iy = grid_axis(p.y, G, h), iz = grid_axis(p.z, G, h);
    int4 nd = T.node[T.grid[((size_t)ix * G + iy) * G + iz]];
    double lx, ly, lz, size;
    if (nd.x < 0) {  // a leaf at depth w <= grid_depth: its box is the depth-w cell above this grid cell
        const int up = T.grid_depth - nd.w;
        size = ldexp(2 * kBounds, -nd.w);  // 202 / 2^w, exact
        lx = -kBounds + (ix >> up) * size;
        ly = -kBounds + (iy >> up) * size;
        lz = -kBounds + (iz >> up) * size;
    } else {
        lx = -kBounds + ix * h;
        ly = -kBounds + iy * h;
        lz = -kBounds + iz * h;
        double hx = lx + h, hy = ly + h, hz = lz + h;
        while (nd.x >= 0) {
            const double mx = (lx + hx) / 2.0, my = (ly + hy) / 2.0, mz = (lz + hz) / 2.0;
            const int cx = !(p.x < mx), cy = !(p.y < my), cz = !(p.z < mz);
            if (cx) lx = mx; else hx = mx;
            if (cy) ly = my; else hy = my;
            if (cz) lz = mz; else hz = mz;
            nd = T.node[nd.x + 4 * cx + 2 * cy + cz];
        }
        size = hx - lx;
    }
    const double q = size / 4;
    const int sx = (p.x >= lx + q) + (p.x >= lx + 2 * q) + (p.x >= lx + 3 * q);
    const int sy = (p.y >= ly + q) + (p.y >= ly + 2 * q) + (p.y >= ly + 3 * q);
    const int sz = (p.z >= lz + q) + (p.z >= lz + 2 * q) + (p.z >= lz + 3 * q);
    subcell = 16 * sx + 4 * sy + sz;
    return nd;
}

// s / P for 0 <= s < 2^22, P > 0, with a float reciprocal and an exact correction step
__device__ __forceinline__ int fast_div(int s, int P, float invP) {
    int q = (int)((float)s * invP);
    if (q * P > s) --q;
    else if ((q + 1) * P <= s) ++q;
    return q;
}

__device__ __forceinline__ bool outside_root(const V3 &p) {
    return p.x < -kBounds || p.x > kBounds || p.y < -kBounds || p.y > kBounds || p.z < -kBounds || p.z > kBounds;
}

// get_target_data's tail (:361-375): barycentric_interpolation on the raw (un-projected) point
__device__ __forceinline__ double emit_sample(const SamplesArgs &a, size_t g, const V3 &p, int t) {
    const TriRec &r = a.tree.rec[t];
    double wa, wb, wc;
    area_weights(rec_v0(r), rec_v1(r), rec_v2(r), p, wa, wb, wc);
    if (a.tval) {
        const double v = wa * a.tfeat[r.id[0]] + wb * a.tfeat[r.id[1]] + wc * a.tfeat[r.id[2]];
        a.tval[g] = v;
        return v;
    } else {
        a.stri[g] = t;
        a.sw3[3 * g] = wa;
        a.sw3[3 * g + 1] = wb;
        a.sw3[3 * g + 2] = wc;
    }
    return 0.0;
}

__device__ __forceinline__ void emit_failure(const SamplesArgs &a, size_t g, int code) {
    raise_status(a.status, code);
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    if (a.tval) {
        a.tval[g] = nan;
    } else {
        a.stri[g] = code;
        a.sw3[3 * g] = a.sw3[3 * g + 1] = a.sw3[3 * g + 2] = nan;
    }
}

// get_sim_for_min (M/similarities.h:48-58) of a moving patch A and a sampled target patch B with weights W,
// evaluated by one wavefront (lane-strided sums + shuffle reduction)
__device__ __forceinline__ double patch_similarity(const double *A, const double *W, const double *B, int P, int lane, int simmeasure) {
    if (simmeasure == 2) {
        // sparsesimkernel::corr, M/similarities.cpp:129-158 (two passes: weighted means, then moments)
        double sw = 0, ma = 0, mb = 0;
        for (int i = lane; i < P; i += 64) {
            sw += W[i];
            ma += W[i] * A[i];
            mb += W[i] * B[i];
        }
        sw = wave_sum(sw);
        ma = wave_sum(ma);
        mb = wave_sum(mb);
        if (sw > 0.0) {
            ma /= sw;
            mb /= sw;
        }
        double pr = 0, va = 0, vb = 0;
        for (int i = lane; i < P; i += 64) {
            const double da = A[i] - ma, db = B[i] - mb;
            pr += W[i] * da * db;
            va += W[i] * da * da;
            vb += W[i] * db * db;
        }
        pr = wave_sum(pr);
        va = wave_sum(va);
        vb = wave_sum(vb);
        if (sw > 0.0) {
            pr /= sw;
            va /= sw;
            vb /= sw;
        }
        const double r = (va == 0.0 || vb == 0.0) ? 0.0 : pr / (sqrt(va) * sqrt(vb));
        return 1 - (1 + r) * 0.5;
    }
    // sparsesimkernel::SSD, M/similarities.cpp:179-188
    double pr = 0;
    for (int i = lane; i < P; i += 64) {
        const double df = A[i] - B[i];
        pr += W[i] * df * df;
    }
    pr = wave_sum(pr);
    return sqrt(pr) / P;
}

}  // namespace

__global__ __launch_bounds__(256) void k_unary_samples(SamplesArgs a) {
    extern __shared__ double lds[];
    double *sx = lds, *sy = sx + a.pmax, *sz = sy + a.pmax;
    double *sA = sz + a.pmax, *sW = sA + a.pmax;                 // moving feature and weights of the patch
    double *sR = sW + a.pmax;                                    // L x 9
    double *sT = sR + 9 * a.L;                                   // L x pmax sampled target values
    unsigned *queue = reinterpret_cast<unsigned *>(sT + (size_t)a.L * a.pmax);  // kQueueCap
    int *nin = reinterpret_cast<int *>(queue + kQueueCap);         // kChunk: containing triangles found
    int *win = nin + kChunk;                                        // kChunk: one of them
    __shared__ int s_qn, s_ndefer;
    const int tid = threadIdx.x, lane = tid & 63;
    const int per = (a.N + 7) >> 3;
    const int slots = 8 * per;

    // blockIdx -> control point: the blocks of one XCD (blockIdx % 8, round-robin dispatch) get a contiguous id
    // range, i.e. spatial neighbours on the icosphere, so each XCD's L2 keeps its own part of the target.
    // (A device-side work queue was measured 2.4x slower here: the returning atomics serialise the blocks.)
    // The labels of a control point are split over a.nsplit workgroups (same XCD) so that one workgroup's samples
    // fit a single LDS pass and the grid has enough workgroups to balance over the 256 CUs.
    {
        const int part = blockIdx.x / slots, slot = blockIdx.x - part * slots;
        const int node = (slot & 7) * per + (slot >> 3);
        if (node >= a.N) return;
        const int lper = (a.L + a.nsplit - 1) / a.nsplit;
        const int l_beg = part * lper, l_end = min(a.L, l_beg + lper);
        if (l_beg >= l_end) return;
        const int beg = a.pptr[node], P = a.pptr[node + 1] - beg;
        const size_t gbase = (size_t)a.L * beg;
        for (int i = tid; i < P; i += 256) {
            const int s = a.pidx[beg + i];
            sx[i] = a.src[s];
            sy[i] = a.src[a.Nsrc + s];
            sz[i] = a.src[2 * a.Nsrc + s];
            if (a.U) {
                sA[i] = a.sfeat[s];
                sW[i] = a.cfw ? a.cfw[s] : 1.0;
            }
        }
        for (int k = tid; k < 9 * a.L; k += 256) sR[k] = a.rnl[(size_t)node * a.L * 9 + k];
        if (tid == 0) s_ndefer = 0;
        const int sbeg = l_beg * P, send = l_end * P;  // this workgroup's samples: s = label * P + patch point
        const float invP = 1.0f / (float)max(P, 1);

        for (int base = sbeg; base < send; base += kChunk) {
            const int nchunk = min(kChunk, send - base);
            for (int k = tid; k < kChunk; k += 256) nin[k] = 0;
            if (tid == 0) s_qn = 0;
            __syncthreads();

            // ---- phase A: per sample (one lane each): rotate, locate the octree leaf, cone-filter its entries.
            // Every lane runs the same number of rounds so that the queue reservation can use wavefront shuffles.
            for (int r0 = 0; r0 < nchunk; r0 += 256) {
                const int sl = r0 + tid;
                unsigned long long pm = 0ull;  // entries that passed the cone filter
                int lbeg = 0;
                bool defer = false;
                if (sl < nchunk) {
                    const int s = base + sl;
                    const int l = fast_div(s, P, invP), i = s - l * P;
                    const V3 p = rotate(sR + 9 * l, mk(sx[i], sy[i], sz[i]));
                    if (outside_root(p)) {
                        nin[sl] = -1;
                    } else {
                        int sub;
                        const int4 leaf = locate_leaf(a.tree, p, sub);
                        lbeg = leaf.y;
                        if (leaf.z < 0) {
                            // no masks: an oversized leaf (the split heuristic refused to split it) needs the complete
                            // search; an empty one yields no pair and reaches the fix-up list through nin == 0
                            defer = (-leaf.x - 1) > 64;
                        } else {
                            const float qx = (float)p.x, qy = (float)p.y, qz = (float)p.z;
                            const float inv = rsqrtf(qx * qx + qy * qy + qz * qz);
                            const float fx = qx * inv, fy = qy * inv, fz = qz * inv;
                            const float4 *cone = a.tree.cone + leaf.y;
                            // entries a point of this sub-cell can hit at all; cone-test those, four loads in flight
                            unsigned long long mm = a.tree.mask[(size_t)leaf.z * 64 + sub];
                            float bestdot = -1.f;
                            int best = -1;
                            while (mm) {
                                int e[4];
                                bool v[4];
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    v[k] = mm != 0ull;
                                    e[k] = v[k] ? __ffsll((long long)mm) - 1 : 0;
                                    mm &= mm - 1ull;  // 0 stays 0
                                }
                                float4 c[4];
#pragma unroll
                                for (int k = 0; k < 4; ++k) c[k] = cone[e[k]];
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    const float dt = fabsf(__builtin_fmaf(c[k].z, fz, __builtin_fmaf(c[k].y, fy, c[k].x * fx)));
                                    if (v[k] && dt >= c[k].w) {
                                        pm |= 1ull << e[k];
                                        if (dt > bestdot) {  // the cone axis is the triangle's centroid direction: nearest centroid
                                            bestdot = dt;
                                            best = e[k];
                                        }
                                    }
                                }
                            }
                            // Fast path on simple surfaces: on a near-regular mesh the nearest centroid's triangle is the
                            // containing one; if the projection is safely inside it, it is the reference's answer.
                            if (a.tree.simple && best >= 0) {
                                const int t = a.tree.leaf_tri[leaf.y + best];
                                if (safely_inside(a.tree.rec[t], p)) {
                                    const double val = emit_sample(a, gbase + s, p, t);
                                    if (a.U) sT[l * a.pmax + i] = val;
                                    nin[sl] = kDone;
                                    pm = 0ull;
                                }
                            }
                        }
                    }
                }
                // reserve queue space: wavefront prefix sum of the per-lane pair counts, one LDS atomic per wave
                const int cntp = __popcll(pm);
                int incl = cntp;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const int up = __shfl_up(incl, off, 64);
                    if (lane >= off) incl += up;
                }
                int wbase = 0;
                if (lane == 63 && incl > 0) wbase = atomicAdd(&s_qn, incl);
                wbase = __shfl(wbase, 63, 64);
                int pos = wbase + incl - cntp;
                if (cntp > 0) {
                    if (pos + cntp > kQueueCap) {
                        defer = true;  // queue full: the fix-up kernel takes this sample; its reserved slots must not stay stale
                        for (int j = pos; j < kQueueCap; ++j) queue[j] = kInvalidPair;
                    } else {
                        const int *lt = a.tree.leaf_tri + lbeg;
                        const unsigned tag = (unsigned)sl << kTriBits;
                        while (pm) {
                            const int e = __ffsll((long long)pm) - 1;
                            pm &= pm - 1ull;
                            queue[pos++] = tag | (unsigned)lt[e];
                        }
                    }
                }
                if (defer) nin[sl] = kDeferred;  // the fix-up kernel handles this sample
            }
            __syncthreads();

#if !(MSM_VARIANT & 1)
            // ---- phase B: exact inside test per (sample, triangle) pair
            const int qn = min(s_qn, kQueueCap);
            for (int j = tid; j < qn; j += 256) {
                const unsigned e = queue[j];
                if (e == kInvalidPair) continue;
                const int sl = (int)(e >> kTriBits), t = (int)(e & ((1u << kTriBits) - 1));
                const int s = base + sl;
                const int l = fast_div(s, P, invP), i = s - l * P;
                const V3 p = rotate(sR + 9 * l, mk(sx[i], sy[i], sz[i]));
                V3 mp;
                if (inside_test(a.tree.rec[t], p, mp)) {
                    win[sl] = t;
                    atomicAdd(&nin[sl], 1);
                }
            }
            __syncthreads();

#endif
#if !(MSM_VARIANT & 3)
            // ---- phase C: interpolate, or defer
            for (int sl = tid; sl < nchunk; sl += 256) {
                const int s = base + sl;
                const int n = nin[sl];
                if (n == 1) {
                    const int l = fast_div(s, P, invP), i = s - l * P;
                    const V3 p = rotate(sR + 9 * l, mk(sx[i], sy[i], sz[i]));
                    const double v = emit_sample(a, gbase + s, p, win[sl]);
                    if (a.U) sT[l * a.pmax + i] = v;
                } else if (n == kDone) {
                    // emitted in phase A
                } else if (n < 0) {
                    emit_failure(a, gbase + s, MSM_ERR_OUTSIDE);
                    atomicAdd(&s_ndefer, 1);
                } else {
                    atomicAdd(&s_ndefer, 1);
                    const unsigned slot = atomicAdd(a.fix_count, 1u);
                    if (slot < a.fix_cap) a.fix_list[slot] = ((unsigned long long)node << 32) | (unsigned)s;
                    else raise_status(a.status, MSM_ERR_CAPACITY);
                }
            }
            __syncthreads();
#endif
        }
        if (tid == 0 && a.nsamples) atomicAdd(a.nsamples, (unsigned long long)(send - sbeg));
        // ---- reduction: every sample of this control point is in LDS unless some were deferred
        if (a.U) {
            __syncthreads();  // s_ndefer and sT are final (also when the patch is empty and the loop never ran)
            if (s_ndefer == 0) {
                const double absw = a.absw[node];
                for (int l = l_beg + (tid >> 6); l < l_end; l += 4) {
                    const double cost = patch_similarity(sA, sW, sT + l * a.pmax, P, lane, a.simmeasure);
                    if (lane == 0) a.U[(size_t)l * a.N + node] = absw * cost;
                }
            } else if (tid == 0) {
                a.redo_list[atomicAdd(a.redo_count, 1u)] = node;
            }
        }
    }
}

// the rare samples: complete reference search (several containing triangles -> dist_to_point tie-break; none ->
// sibling leaves, then nearest vertex)
__global__ __launch_bounds__(256) void k_unary_fixup(SamplesArgs a) {
    const unsigned n = min(*a.fix_count, a.fix_cap);
    for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        const unsigned long long e = a.fix_list[j];
        const int node = (int)(e >> 32), s = (int)(e & 0xffffffffu);
        const int beg = a.pptr[node], P = a.pptr[node + 1] - beg;
        const int l = s / P, i = s - l * P;
        const int sv = a.pidx[beg + i];
        const V3 p = rotate(a.rnl + ((size_t)node * a.L + l) * 9, mk(a.src[sv], a.src[a.Nsrc + sv], a.src[2 * a.Nsrc + sv]));
        const size_t g = (size_t)a.L * beg + s;
        const int t = find_closest_triangle(a.tree, p);
        if (t < 0) emit_failure(a, g, t);
        else emit_sample(a, g, p, t);
    }
}

// ------------------------------------------------------------------------------------------------
// Univariate reduction: AbsoluteWeights[node] * get_sim_for_min(source patch, target patch, weights)
// (UnivariateNonLinearSRegDiscreteCostFunction::computeUnaryCost, :378-383)
// ------------------------------------------------------------------------------------------------
struct ReduceArgs {
    int N, L, Nsrc;
    const double *sfeat;  // D x Nsrc
    const double *cfw;    // rows x Nsrc or nullptr
    int cfw_rows;
    const int *pptr, *pidx;
    const double *absw;
    const double *tval;
    int pmax;
    int simmeasure;
    double *U;
    const int *redo_list;            // nodes to reduce (nullptr: all N)
    const unsigned int *redo_count;
};

__global__ __launch_bounds__(256) void k_unary_reduce_univariate(ReduceArgs a) {
    extern __shared__ double lds[];
    double *sA = lds, *sW = sA + a.pmax;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned n = a.redo_count ? *a.redo_count : (unsigned)a.N;
    for (unsigned k = blockIdx.x; k < n; k += gridDim.x) {
        const int node = a.redo_list ? a.redo_list[k] : (int)k;
        const int beg = a.pptr[node], P = a.pptr[node + 1] - beg;
        __syncthreads();
        for (int i = tid; i < P; i += 256) {
            const int s = a.pidx[beg + i];
            sA[i] = a.sfeat[s];
            sW[i] = (a.cfw && a.cfw_rows >= 1) ? a.cfw[s] : 1.0;
        }
        __syncthreads();
        const double absw = a.absw[node];
        for (int l = wave; l < a.L; l += 4) {
            const double cost = patch_similarity(sA, sW, a.tval + (size_t)a.L * beg + (size_t)l * P, P, lane, a.simmeasure);
            if (lane == 0) a.U[(size_t)l * a.N + node] = absw * cost;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Multivariate / patchwise reductions from the stored (triangle, raw weights) of every sample.
//   multivariate (M/DiscreteCostFunction.cpp:444-458): mean over the patch points of a D-long feature-vector
//                similarity; one lane per patch point, the D loop runs inside the lane;
//   patchwise    (:680-692): per feature channel a patch similarity (lanes over points, shuffle reduction),
//                averaged over the channels.
// One workgroup per control point, one wavefront per label.
// ------------------------------------------------------------------------------------------------
struct ReduceMvArgs {
    int N, L, Nsrc, D;
    const double *tfeat;  // V x D vertex-major
    const TriRec *rec;
    const double *sfeat;  // D x Nsrc
    const double *cfw;
    int cfw_rows;
    const int *pptr, *pidx;
    const double *absw;
    const int *stri;
    const double *sw3;
    int simmeasure;
    int patchwise;
    double *U;
};

__global__ __launch_bounds__(256) void k_unary_reduce_features(ReduceMvArgs a) {
    const int node = blockIdx.x;
    const int beg = a.pptr[node], P = a.pptr[node + 1] - beg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double absw = a.absw[node];
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    for (int l = wave; l < a.L; l += 4) {
        const size_t g0 = (size_t)a.L * beg + (size_t)l * P;
        double cost;
        if (!a.patchwise) {
            double acc = 0.0;
            bool bad = false;
            for (int i = lane; i < P; i += 64) {
                const int t = a.stri[g0 + i];
                if (t < 0) {
                    bad = true;
                    continue;
                }
                const TriRec &r = a.rec[t];
                const double *f0 = a.tfeat + (size_t)r.id[0] * a.D, *f1 = a.tfeat + (size_t)r.id[1] * a.D, *f2 = a.tfeat + (size_t)r.id[2] * a.D;
                acc += feature_vector_similarity(a.simmeasure, a.sfeat, a.cfw, a.cfw_rows, a.Nsrc, a.pidx[beg + i], a.D, f0, f1, f2,
                                                 a.sw3[3 * (g0 + i)], a.sw3[3 * (g0 + i) + 1], a.sw3[3 * (g0 + i) + 2]);
            }
            acc = wave_sum(acc);
            cost = P > 0 ? acc / P : acc;
            if (__ballot(bad)) cost = nan;
        } else {
            double total = 0.0;
            bool bad = false;
            for (int d = 0; d < a.D; ++d) {
                const double *A = a.sfeat + (size_t)d * a.Nsrc;
                auto Bv = [&](int i) {
                    const int t = a.stri[g0 + i];
                    if (t < 0) {
                        bad = true;
                        return nan;
                    }
                    const TriRec &r = a.rec[t];
                    return a.sw3[3 * (g0 + i)] * a.tfeat[(size_t)r.id[0] * a.D + d] + a.sw3[3 * (g0 + i) + 1] * a.tfeat[(size_t)r.id[1] * a.D + d] +
                           a.sw3[3 * (g0 + i) + 2] * a.tfeat[(size_t)r.id[2] * a.D + d];
                };
                auto Wv = [&](int i) { return (a.cfw && a.cfw_rows >= 1) ? a.cfw[a.pidx[beg + i]] : 1.0; };
                double c;
                if (a.simmeasure == 2) {
                    double sw = 0, ma = 0, mb = 0;
                    for (int i = lane; i < P; i += 64) {
                        const double w = Wv(i);
                        sw += w;
                        ma += w * A[a.pidx[beg + i]];
                        mb += w * Bv(i);
                    }
                    sw = wave_sum(sw);
                    ma = wave_sum(ma);
                    mb = wave_sum(mb);
                    if (sw > 0.0) {
                        ma /= sw;
                        mb /= sw;
                    }
                    double pr = 0, va = 0, vb = 0;
                    for (int i = lane; i < P; i += 64) {
                        const double w = Wv(i), da = A[a.pidx[beg + i]] - ma, db = Bv(i) - mb;
                        pr += w * da * db;
                        va += w * da * da;
                        vb += w * db * db;
                    }
                    pr = wave_sum(pr);
                    va = wave_sum(va);
                    vb = wave_sum(vb);
                    if (sw > 0.0) {
                        pr /= sw;
                        va /= sw;
                        vb /= sw;
                    }
                    const double r = (va == 0.0 || vb == 0.0) ? 0.0 : pr / (sqrt(va) * sqrt(vb));
                    c = 1 - (1 + r) * 0.5;
                } else {
                    double pr = 0;
                    for (int i = lane; i < P; i += 64) {
                        const double df = A[a.pidx[beg + i]] - Bv(i);
                        pr += Wv(i) * df * df;
                    }
                    pr = wave_sum(pr);
                    c = sqrt(pr) / P;
                }
                total += c;
            }
            cost = total / a.D;
            if (__ballot(bad)) cost = nan;
        }
        if (lane == 0) a.U[(size_t)l * a.N + node] = absw * cost;
    }
}

// ------------------------------------------------------------------------------------------------
// launch
// ------------------------------------------------------------------------------------------------
static size_t samples_lds(int pmax, int L) {
    return sizeof(double) * (5 * (size_t)pmax + 9 * (size_t)L + (size_t)L * pmax) + sizeof(unsigned) * kQueueCap + sizeof(int) * 2 * kChunk;
}

static int launch_samples(msm_ctx *ctx, const UnaryLaunch &u, const UnaryWeightsScratch *w, SamplesArgs &a) {
    if (u.tree.nnodes <= 0 || !u.tree.mask) return fail(MSM_ERR_STATE, "target search structure missing");
    a.tree = u.tree;
    a.tfeat = u.tfeat;
    a.N = u.N;
    a.L = u.L;
    a.rnl = u.rnl;
    a.src = u.src;
    a.Nsrc = u.Nsrc;
    a.pptr = u.pptr;
    a.pidx = u.pidx;
    a.pmax = u.pmax;
    a.tval = w ? nullptr : u.tval;
    a.stri = w ? w->stri : nullptr;
    a.sw3 = w ? w->sw3 : nullptr;
    a.fix_list = u.fix_list;
    a.fix_count = u.fix_count;
    a.fix_cap = u.fix_cap;
    a.sfeat = u.sfeat;
    a.cfw = (u.cfw && u.cfw_rows >= 1) ? u.cfw : nullptr;
    a.absw = u.absw;
    a.simmeasure = u.simmeasure;
    a.U = w ? nullptr : u.U;  // the fused reduction is the univariate one
    a.redo_list = u.redo_list;
    a.redo_count = u.fix_count + 1;
    a.status = ctx->d_status;
    a.nsamples = u.nsamples;
    if (u.ntri >= (1 << kTriBits)) return fail(MSM_ERR_CAPACITY, "target mesh has %d triangles; the sample queue packs ids in %d bits", u.ntri, kTriBits);
    const size_t lds = samples_lds(u.pmax, u.L);
    if (lds > 64 * 1024) {
        if (lds > 160 * 1024) return fail(MSM_ERR_CAPACITY, "patch of %d points x %d labels does not fit in LDS", u.pmax, u.L);
        MSM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_unary_samples), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    MSM_HIP(hipMemsetAsync(u.fix_count, 0, 2 * sizeof(unsigned), ctx->stream));  // fix-up and redo counters
    a.nsplit = std::max(1, std::min(4, (int)(((size_t)u.L * u.pmax + kChunk - 1) / kChunk)));
    const int blocks = a.nsplit * 8 * ((u.N + 7) / 8);
    if (u.ev_start) MSM_HIP(hipEventRecord(u.ev_start, ctx->stream));
    hipLaunchKernelGGL(k_unary_samples, dim3(blocks), dim3(256), lds, ctx->stream, a);
    MSM_HIP(hipGetLastError());
    if (u.ev_stop) MSM_HIP(hipEventRecord(u.ev_stop, ctx->stream));
    hipLaunchKernelGGL(k_unary_fixup, dim3(64), dim3(256), 0, ctx->stream, a);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_unary_univariate(msm_ctx *ctx, const UnaryLaunch &u) {
    SamplesArgs a;
    int st = launch_samples(ctx, u, nullptr, a);
    if (st) return st;
    // control points that had deferred samples are reduced now that the fix-up kernel has filled them in
    ReduceArgs r;
    r.N = u.N;
    r.L = u.L;
    r.Nsrc = u.Nsrc;
    r.sfeat = u.sfeat;
    r.cfw = u.cfw;
    r.cfw_rows = u.cfw_rows;
    r.pptr = u.pptr;
    r.pidx = u.pidx;
    r.absw = u.absw;
    r.tval = u.tval;
    r.pmax = u.pmax;
    r.simmeasure = u.simmeasure;
    r.U = u.U;
    r.redo_list = u.redo_list;
    r.redo_count = u.fix_count + 1;
    hipLaunchKernelGGL(k_unary_reduce_univariate, dim3(64), dim3(256), sizeof(double) * 2 * (size_t)u.pmax, ctx->stream, r);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_unary_multivariate(msm_ctx *ctx, const UnaryLaunch &u, const UnaryWeightsScratch &w, bool patchwise) {
    SamplesArgs a;
    int st = launch_samples(ctx, u, &w, a);
    if (st) return st;
    ReduceMvArgs r;
    r.N = u.N;
    r.L = u.L;
    r.Nsrc = u.Nsrc;
    r.D = u.D;
    r.tfeat = u.tfeat;
    r.rec = u.tree.rec;
    r.sfeat = u.sfeat;
    r.cfw = u.cfw;
    r.cfw_rows = u.cfw_rows;
    r.pptr = u.pptr;
    r.pidx = u.pidx;
    r.absw = u.absw;
    r.stri = w.stri;
    r.sw3 = w.sw3;
    r.simmeasure = u.simmeasure;
    r.patchwise = patchwise ? 1 : 0;
    r.U = u.U;
    hipLaunchKernelGGL(k_unary_reduce_features, dim3(u.N), dim3(256), 0, ctx->stream, r);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

}  // namespace msm
