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
#include <cstdlib>

#include "kernels.hpp"
#include "search_device.hpp"
#include "similarity_device.hpp"

#ifndef MSM_VARIANT
#define MSM_VARIANT 0  // profiling only: 1 = stop after phase A, 2 = stop after phase B, 4 = skip the cone filter
#endif

namespace msm {

namespace {

constexpr int kFixSegs = 64;       // segments of the fix-up list (see SamplesArgs)
constexpr int kCntStride = 32;     // counters sit 128 bytes apart
constexpr int kChunk = 768;       // samples per LDS pass (3 rounds of 256 lanes)
constexpr int kQueueCap = 3072;   // (sample, triangle) pairs per pass
constexpr int kDeferred = 1 << 20; // nin marker: leave this sample to the fix-up kernel
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
    const double *tfeat;   // target features, V x tD vertex-major (the univariate class reads row 1 = column 0 only,
    int tD;                // M/DiscreteCostFunction.cpp:371), or nullptr when only weights are wanted
    int N, L;
    const double *rnl;     // N x L x 9
    const double *src;     // 3 x Nsrc SoA
    int Nsrc;
    const int *pptr, *pidx;
    const int *order;      // launch order of the control points
    int pmax;
    int nsplit;            // workgroups per control point (label ranges)
    // outputs, indexed by global sample id g = L*pptr[node] + l*P + i
    double *tval;          // D == 1: interpolated reference feature
    int *stri;             // otherwise: triangle id and the three raw barycentric weights
    double *sw3;           // 3 doubles per sample
    // Fix-up list: kFixSegs segments (a workgroup appends to segment blockIdx % kFixSegs, which is sized for all
    // samples of its workgroups: it cannot overflow).  One list with one counter would funnel every append of
    // the chip through a single address; measured: 10 ns per append, i.e. most of the kernel's time.
    unsigned long long *fix_list;  // global sample id (L * pptr[node] + local sample)
    double *fix_pt;                // the rotated point of the listed sample, 3 doubles per slot: the fix-up kernel starts from it
    unsigned int *fix_cnt;         // counter of segment s at fix_cnt[kCntStride * (1 + s)]; [0] is the redo counter
    const unsigned int *fix_off;   // kFixSegs + 1 segment offsets into fix_list
    // fused reduction (univariate): moving feature, weights, AbsoluteWeights, output table
    const double *sfeat;   // Nsrc (feature row 1)
    const double *cfw;     // Nsrc (weight row 1) or nullptr
    const double *absw;    // N
    int simmeasure;
    double percentile;     // DICE measures
    double *U;             // L x N
    int *redo_list;        // nodes whose reduction must wait for the fix-up kernel
    unsigned int *redo_count;
    int *status;
};

// s / P for 0 <= s < 2^22, P > 0, with a float reciprocal and an exact correction step
__device__ __forceinline__ int fast_div(int s, int P, float invP) {
    int q = (int)((float)s * invP);
    if (q * P > s) --q;
    else if ((q + 1) * P <= s) ++q;
    return q;
}

// get_target_data's tail (:361-375): barycentric_interpolation on the raw (un-projected) point
__device__ __forceinline__ double emit_sample(const SamplesArgs &a, size_t g, const V3 &p, int t) {
    const TriRec &r = a.tree.rec[t];
    double wa, wb, wc;
    area_weights(rec_v0(r), rec_v1(r), rec_v2(r), p, wa, wb, wc);
    if (a.tval) {
        const double v = wa * a.tfeat[(size_t)r.id[0] * a.tD] + wb * a.tfeat[(size_t)r.id[1] * a.tD] + wc * a.tfeat[(size_t)r.id[2] * a.tD];
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
__device__ __forceinline__ double patch_similarity(const double *A, const double *W, const double *B, int P, int lane, int simmeasure, double percentile) {
    if (simmeasure == 4 || simmeasure == 5) return patch_dice(A, B, P, lane, simmeasure, percentile);
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
    extern __shared__ __align__(16) double lds[];
    double *sx = lds, *sy = sx + a.pmax, *sz = sy + a.pmax;
    double *sA = sz + a.pmax, *sW = sA + a.pmax;                 // moving feature and weights of the patch
    double *sR = sW + a.pmax;                                    // L x 9
    const int lper_all = (a.L + a.nsplit - 1) / a.nsplit;
    double *sT = sR + 9 * a.L;                                   // (labels of this workgroup) x pmax sampled target values
    unsigned *queue = reinterpret_cast<unsigned *>(sT + (size_t)lper_all * a.pmax);  // kQueueCap
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
        const int at = (slot & 7) * per + (slot >> 3);
        if (at >= a.N) return;
        const int node = a.order[at];
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
                                    if (v[k] && dt >= c[k].w) pm |= 1ull << e[k];
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
                    if (a.U) sT[(l - l_beg) * a.pmax + i] = v;
                } else if (n < 0) {
                    emit_failure(a, gbase + s, MSM_ERR_OUTSIDE);
                    atomicAdd(&s_ndefer, 1);
                } else {
                    atomicAdd(&s_ndefer, 1);
                    const int seg = blockIdx.x % kFixSegs;
                    const unsigned slot = a.fix_off[seg] + atomicAdd(&a.fix_cnt[kCntStride * (1 + seg)], 1u);
                    if (slot < a.fix_off[seg + 1]) {
                        const int l = fast_div(s, P, invP), i = s - l * P;
                        const V3 p = rotate(sR + 9 * l, mk(sx[i], sy[i], sz[i]));
                        a.fix_list[slot] = gbase + (unsigned long long)s;
                        a.fix_pt[3 * (size_t)slot] = p.x;
                        a.fix_pt[3 * (size_t)slot + 1] = p.y;
                        a.fix_pt[3 * (size_t)slot + 2] = p.z;
                    } else {
                        raise_status(a.status, MSM_ERR_CAPACITY);
                    }
                }
            }
            __syncthreads();
#endif
        }
        // ---- reduction: every sample of this control point is in LDS unless some were deferred
        if (a.U) {
            __syncthreads();  // s_ndefer and sT are final (also when the patch is empty and the loop never ran)
            if (s_ndefer == 0) {
                const double absw = a.absw[node];
                for (int l = l_beg + (tid >> 6); l < l_end; l += 4) {
                    const double cost = patch_similarity(sA, sW, sT + (l - l_beg) * a.pmax, P, lane, a.simmeasure, a.percentile);
                    if (lane == 0) a.U[(size_t)l * a.N + node] = absw * cost;
                }
            } else if (tid == 0) {
                a.redo_list[atomicAdd(a.redo_count, 1u)] = node;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Simple-surface targets: the ray table (search_device.hpp, octree.cpp: build_ray_table) settles a sample with one
// direction-cell load and ~1.3 candidate records; what it cannot settle (about 2 % on an icosphere target) goes to
// the fix-up list, where k_unary_fixup runs the complete search.  Same workgroup -> control point mapping as
// k_unary_samples; after staging every wavefront strides over the samples on its own.
//
// The kernel is bound by the vector L1's line-lookup rate (one 128-byte line per cycle per CU: a load whose 64
// lanes hit 64 different lines costs 64 cycles, whatever its width), not by HBM or the ALUs: about 13 line
// lookups per sample (cell 1, edge planes 3, vertices + features 6, retries) = 4.1e7 per table, 161 k cycles per CU.
// Two variants in which the wavefront fetches the 64 records as a team (lane j of load k reads piece (64k+j) % 9 of
// the record of lane (64k+j) / 9, straight into LDS with global_load_lds_dwordx4) halved the lookups, but at 9 KB of
// LDS per wavefront they cut the occupancy to 3 waves per SIMD and lengthen the dependent chain (shuffle -> load ->
// LDS -> test): 123 us (all four candidate rounds as a team) and 107 us (first candidate only) against 85 us for
// the plain per-lane loads below.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_unary_rays(SamplesArgs a) {
    extern __shared__ __align__(16) double lds[];
    double *sx = lds, *sy = sx + a.pmax, *sz = sy + a.pmax, *sR = sz + a.pmax;
    const int tid = threadIdx.x, lane = tid & 63;
    const int per = (a.N + 7) >> 3;
    const int slots = 8 * per;
    const int part = blockIdx.x / slots, slot = blockIdx.x - part * slots;
    const int at = (slot & 7) * per + (slot >> 3);
    if (at >= a.N) return;
    const int node = a.order[at];
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
    }
    for (int k = 9 * l_beg + tid; k < 9 * l_end; k += 256) sR[k] = a.rnl[(size_t)node * a.L * 9 + k];
    const int sbeg = l_beg * P, send = l_end * P;
    const float invP = 1.0f / (float)max(P, 1);
    // unsettled samples are collected in LDS and appended to the global list with ONE atomic per workgroup
    // (a returning atomic per wavefront on one address was measured to triple the kernel's time)
    int *sleft = reinterpret_cast<int *>(sR + 9 * a.L);
    __shared__ int s_nleft;
    __shared__ unsigned s_base;
    if (tid == 0) s_nleft = 0;
    __syncthreads();
    for (int s0 = sbeg; s0 < send; s0 += 256) {
        const int s = s0 + tid;
        const bool act = s < send;
        bool done = false;
        V3 p = mk(0.0, 0.0, 0.0);
        float fx = 0.f, fy = 0.f, fz = 0.f;
        int4 c = make_int4(-1, -1, -1, -1);
        if (act) {
            const int l = fast_div(s, P, invP), i = s - l * P;
            p = rotate(sR + 9 * l, mk(sx[i], sy[i], sz[i]));
            c = ray_cell_of(a.tree, p, fx, fy, fz);
        }
        if (act) {
            if (c.x >= 0) {
                // the first candidate is the answer three times out of four: its vertices are requested together with
                // its edge planes, so a settled sample costs two dependent loads after the cell
                const float4 *rec = a.tree.ray_tri + (size_t)kRayPieces * c.x;
                const float4 e0 = rec[0], e1 = rec[1], e2 = rec[2];
                const double2 *dv = reinterpret_cast<const double2 *>(rec + 3);
                double2 d0 = dv[0], d1 = dv[1], d2 = dv[2], d3 = dv[3], d4 = dv[4], d5 = dv[5];
                int t = c.x;
                bool vouch_needed = __float_as_int(e1.w) >= 0;
                float4 ev = e1;
                if (!ray_accepts(e0, e1, e2, fx, fy, fz)) {
                    t = -1;
                    int4 mo = make_int4(c.w, -1, -1, -1);
                    if (c.w < -1) mo = a.tree.ray_more[-2 - c.w];  // a cell with more than four candidates
                    // a real loop, not unrolled: this path is taken by one lane in four and must not cost registers
#pragma unroll 1
                    for (int k = 0; k < 6 && t < 0; ++k) {
                        const int ck = k == 0 ? c.y : (k == 1 ? c.z : (k == 2 ? mo.x : (k == 3 ? mo.y : (k == 4 ? mo.z : mo.w))));
                        if (ck < 0) break;
                        const float4 *r2 = a.tree.ray_tri + (size_t)kRayPieces * ck;
                        const float4 g0 = r2[0], g1 = r2[1], g2 = r2[2];
                        if (ray_accepts(g0, g1, g2, fx, fy, fz)) {
                            t = ck;
                            ev = g1;
                            vouch_needed = __float_as_int(g1.w) >= 0;
                        }
                    }
                    if (t >= 0) {
                        dv = reinterpret_cast<const double2 *>(a.tree.ray_tri + (size_t)kRayPieces * t + 3);
                        d0 = dv[0], d1 = dv[1], d2 = dv[2], d3 = dv[3], d4 = dv[4], d5 = dv[5];
                    }
                }
                if (t >= 0 && vouch_needed && !ray_vouches(a.tree, ev, p)) t = -1;  // its leaf may not list the triangle
                if (t >= 0) {
                    double wa, wb, wc;
                    area_weights(mk(d0.x, d0.y, d1.x), mk(d1.y, d2.x, d2.y), mk(d3.x, d3.y, d4.x), p, wa, wb, wc);
                    const size_t g = gbase + s;
                    if (a.tval) {
                        a.tval[g] = wa * d4.y + wb * d5.x + wc * d5.y;
                    } else {
                        a.stri[g] = t;
                        a.sw3[3 * g] = wa;
                        a.sw3[3 * g + 1] = wb;
                        a.sw3[3 * g + 2] = wc;
                    }
                    done = true;
                }
            }
        }
        const bool left = act && !done;
        const unsigned long long bal = __ballot(left);
        if (bal) {
            const int leader = __ffsll((long long)bal) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(&s_nleft, __popcll(bal));
            base = __shfl(base, leader, 64);
            if (left) sleft[base + __popcll(bal & ((1ull << lane) - 1ull))] = s;
        }
    }
    __syncthreads();
    const int nleft = s_nleft;
    if (nleft > 0) {
        const int seg = blockIdx.x % kFixSegs;
        if (tid == 0) s_base = a.fix_off[seg] + atomicAdd(&a.fix_cnt[kCntStride * (1 + seg)], (unsigned)nleft);
        __syncthreads();
        const unsigned base = s_base, end = a.fix_off[seg + 1];
        for (int j = tid; j < nleft; j += 256) {
            if (base + j < end) {
                const int s = sleft[j];
                const int l = fast_div(s, P, invP), i = s - l * P;
                const V3 p = rotate(sR + 9 * l, mk(sx[i], sy[i], sz[i]));  // as in the loop above: the fix-up kernel need not look anything up
                a.fix_list[base + j] = gbase + (unsigned long long)s;
                a.fix_pt[3 * (size_t)(base + j)] = p.x;
                a.fix_pt[3 * (size_t)(base + j) + 1] = p.y;
                a.fix_pt[3 * (size_t)(base + j) + 2] = p.z;
            } else {
                raise_status(a.status, MSM_ERR_CAPACITY);
            }
        }
    }
}

// The listed samples, eight lanes each.  The list is short (a few per cent of the samples at most), so what
// counts is the length of one wavefront's dependent chain, not throughput: with a lane per sample the exact
// tests of 64 lanes end up at 64 different loop positions and run one after the other (32 us for 2 % of an ico6
// table).  Here the eight lanes of a group share the point and split the leaf's entries (entry e goes to lane
// e % 8): sub-cell mask bit, cone test, exact inside test -- every lane at most 8 times, and the tests of one round
// run together.  Exactly one containing triangle = the reference's answer (R/octree.cpp:166-178: a single passing
// triangle wins whatever its distance); anything else (tie-break by dist_to_point, sibling leaves, nearest vertex,
// oversized leaves) is redone by the group's first lane with the complete search of search_device.hpp (with the
// exclusion boxes and overflow candidates of the ray table these are a handful per table).
__global__ __launch_bounds__(256) void k_unary_fixup(SamplesArgs a) {
    // dense index over the segments: prefix sums of the segment counts (one wavefront, kFixSegs == 64)
    __shared__ unsigned s_pre[kFixSegs + 1];
    if (threadIdx.x < kFixSegs) {
        const int seg = threadIdx.x;
        const unsigned cnt = min(a.fix_cnt[kCntStride * (1 + seg)], a.fix_off[seg + 1] - a.fix_off[seg]);
        unsigned incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned up = __shfl_up(incl, off, 64);
            if (seg >= off) incl += up;
        }
        s_pre[seg + 1] = incl;
        if (seg == 0) s_pre[0] = 0;
    }
    __syncthreads();
    const unsigned n = s_pre[kFixSegs];
    const int lane = threadIdx.x & 63, sub = lane & 7, grp = lane >> 3;
    const unsigned per_block = 256 / 8, stride = gridDim.x * per_block;
    // wavefront-uniform loop: the ballots below need all 64 lanes
    for (unsigned j0 = blockIdx.x * per_block + (threadIdx.x >> 6) * 8; j0 < n; j0 += stride) {
        const unsigned j = j0 + grp;
        const bool valid = j < n;
        V3 p = mk(0.0, 0.0, 0.0);
        size_t g = 0;
        if (valid) {
            int seg = 0;
#pragma unroll
            for (int step = kFixSegs / 2; step > 0; step >>= 1)
                if (s_pre[seg + step] <= j) seg += step;
            const size_t slot = a.fix_off[seg] + (j - s_pre[seg]);
            g = (size_t)a.fix_list[slot];
            p = mk(a.fix_pt[3 * slot], a.fix_pt[3 * slot + 1], a.fix_pt[3 * slot + 2]);
        }
        const int found = group8_find(a.tree, valid, p, lane);
        if (valid && sub == 0) {
            const int t = found == kGroupUndecided ? find_closest_triangle(a.tree, p) : found;  // nothing in the leaf, no masks, outside the root
            if (t < 0) emit_failure(a, g, t);
            else emit_sample(a, g, p, t);
        }
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
    double percentile;
    double *U;
    const int *redo_list;            // nodes to reduce (nullptr: all N)
    const unsigned int *redo_count;
};

__global__ __launch_bounds__(256) void k_unary_reduce_univariate(ReduceArgs a) {
    extern __shared__ __align__(16) double lds[];
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
            const double cost = patch_similarity(sA, sW, a.tval + (size_t)a.L * beg + (size_t)l * P, P, lane, a.simmeasure, a.percentile);
            if (lane == 0) a.U[(size_t)l * a.N + node] = absw * cost;
        }
    }
}

// The same reduction for a whole table whose samples all sit in tval (ray-table path): one workgroup per control
// point stages the moving patch (feature, weight) in LDS once; an 8-lane group per label (32 labels per pass) reads
// that label's sampled target patch, contiguous in tval, into registers while the staging loads are still in
// flight -- the kernel is a chain of dependent loads, so they are issued as early as they can be.  The 8-lane sums
// are DPP row operations.  The weighted mean and variance of the moving patch do not depend on the label and are
// computed once.  The last kernel of a ray-table launch also clears the fix-up counters for the next one.
constexpr int kRedLanes = 8;      // lanes per label
constexpr int kRedKeep = 12;      // target values per lane kept in registers (patches up to 96 points)

__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int off = kRedLanes / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kRedLanes);
    return v;
}

__global__ __launch_bounds__(256) void k_unary_reduce_flat(ReduceArgs a, unsigned int *clear, int clear_words) {
    extern __shared__ __align__(16) double lds[];
    double *sA = lds, *sW = sA + a.pmax;
    __shared__ double s_stat[3];  // sum of weights, weighted mean of A, weighted variance sum of A
    const int tid = threadIdx.x, sub = tid & (kRedLanes - 1), grp = tid / kRedLanes;
    if (blockIdx.x == 0 && clear)
        for (int k = tid; k < clear_words; k += 256) clear[k] = 0u;
    const int node = a.redo_list ? a.redo_list[blockIdx.x] : (int)blockIdx.x;  // launch order = Morton order
    const int beg = a.pptr[node], P = a.pptr[node + 1] - beg;
    const bool has_w = a.cfw && a.cfw_rows >= 1;
    const bool in_regs = true;  // the first kRedKeep values per lane always travel through registers
    double breg[kRedKeep];
#pragma unroll
    for (int k = 0; k < kRedKeep; ++k) breg[k] = 0.0;
    if (grp < a.L) {
        const double *B = a.tval + (size_t)a.L * beg + (size_t)grp * P;
#pragma unroll
        for (int k = 0; k < kRedKeep; ++k) breg[k] = (sub + kRedLanes * k < P) ? B[sub + kRedLanes * k] : 0.0;
    }
    for (int i = tid; i < P; i += 256) {
        const int s = a.pidx[beg + i];
        sA[i] = a.sfeat[s];
        sW[i] = has_w ? a.cfw[s] : 1.0;
    }
    __syncthreads();
    if (a.simmeasure == 2) {
        if (tid < 64) {  // one wavefront: moving-patch statistics
            double sw = 0, ma = 0;
            for (int i = tid; i < P; i += 64) {
                sw += sW[i];
                ma += sW[i] * sA[i];
            }
            sw = wave_sum(sw);
            ma = wave_sum(ma);
            if (sw > 0.0) ma /= sw;
            double va = 0;
            for (int i = tid; i < P; i += 64) {
                const double da = sA[i] - ma;
                va += sW[i] * da * da;
            }
            va = wave_sum(va);
            if (tid == 0) s_stat[0] = sw, s_stat[1] = ma, s_stat[2] = va;
        }
        __syncthreads();
    }
    const double absw = a.absw[node];
    for (int l = grp; l < a.L; l += 256 / kRedLanes) {
        const double *B = a.tval + (size_t)a.L * beg + (size_t)l * P;
        if (!(in_regs && l == grp)) {  // not prefetched: more than 32 labels, or a patch too big for the registers
#pragma unroll
            for (int k = 0; k < kRedKeep; ++k) breg[k] = (sub + kRedLanes * k < P) ? B[sub + kRedLanes * k] : 0.0;
        }
        double cost;
        if (a.simmeasure == 2) {  // sparsesimkernel::corr, M/similarities.cpp:129-158
            const double sw = s_stat[0], ma = s_stat[1];
            double mb = 0;
#pragma unroll
            for (int k = 0; k < kRedKeep; ++k) {
                const int i = sub + kRedLanes * k;
                if (i < P) mb += sW[i] * breg[k];
            }
            for (int i = sub + kRedLanes * kRedKeep; i < P; i += kRedLanes) mb += sW[i] * B[i];
            mb = group_sum(mb);
            if (sw > 0.0) mb /= sw;
            double pr = 0, vb = 0;
#pragma unroll
            for (int k = 0; k < kRedKeep; ++k) {
                const int i = sub + kRedLanes * k;
                if (i < P) {
                    const double da = sA[i] - ma, db = breg[k] - mb;
                    pr += sW[i] * da * db;
                    vb += sW[i] * db * db;
                }
            }
            for (int i = sub + kRedLanes * kRedKeep; i < P; i += kRedLanes) {
                const double da = sA[i] - ma, db = B[i] - mb;
                pr += sW[i] * da * db;
                vb += sW[i] * db * db;
            }
            pr = group_sum(pr);
            vb = group_sum(vb);
            double va = s_stat[2];
            if (sw > 0.0) {
                pr /= sw;
                va /= sw;
                vb /= sw;
            }
            const double r = (va == 0.0 || vb == 0.0) ? 0.0 : pr / (sqrt(va) * sqrt(vb));
            cost = 1 - (1 + r) * 0.5;
        } else {  // sparsesimkernel::SSD, M/similarities.cpp:179-188
            double pr = 0;
#pragma unroll
            for (int k = 0; k < kRedKeep; ++k) {
                const int i = sub + kRedLanes * k;
                if (i < P) {
                    const double df = sA[i] - breg[k];
                    pr += sW[i] * df * df;
                }
            }
            for (int i = sub + kRedLanes * kRedKeep; i < P; i += kRedLanes) {
                const double df = sA[i] - B[i];
                pr += sW[i] * df * df;
            }
            pr = group_sum(pr);
            cost = sqrt(pr) / P;
        }
        if (sub == 0) a.U[(size_t)l * a.N + node] = absw * cost;
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
    double percentile;
    int patchwise;
    int pmax;
    double *U;
};

__global__ __launch_bounds__(256) void k_unary_reduce_features(ReduceMvArgs a) {
    const int node = blockIdx.x;
    const int beg = a.pptr[node], P = a.pptr[node + 1] - beg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double absw = a.absw[node];
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const int nwaves = blockDim.x >> 6;  // 4, or 1 when a patch is so large that only one wavefront's LDS slice fits
    for (int l = wave; l < a.L; l += nwaves) {
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
                acc += feature_vector_similarity(a.simmeasure, a.percentile, a.sfeat, a.cfw, a.cfw_rows, a.Nsrc, a.pidx[beg + i], a.D, f0, f1, f2,
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
                if (a.simmeasure == 4 || a.simmeasure == 5) {
                    // DICE ranks every value against every other one: stage the two patches of this channel in the
                    // wavefront's LDS slice first (the wavefront runs in lockstep: its own LDS writes are visible to all
                    // of its lanes once they have been issued and waited for)
                    extern __shared__ __align__(16) double lds[];
                    double *pa = lds + (size_t)wave * 2 * a.pmax, *pb = pa + a.pmax;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    for (int i = lane; i < P; i += 64) {
                        pa[i] = A[a.pidx[beg + i]];
                        pb[i] = Bv(i);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    c = patch_dice(pa, pb, P, lane, a.simmeasure, a.percentile);
                } else if (a.simmeasure == 2) {
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
// Multivariate reduction, D <= 64, SSD / correlation (M/DiscreteCostFunction.cpp:444-458): eight lanes per patch point.
// With a lane per point and the D loop inside the lane (k_unary_reduce_features) every one of the ~4 D loads of a
// point is its own cache-line lookup (moving features D x Nsrc row-major: a different line per dimension) -- about 250
// lookups per sample, 1.9 ms per table at D = 32.  Here lane j of a group owns the dimensions j, j + 8, ...: the
// group reads the three target rows (vertex-major) and the moving row (a vertex-major copy) as contiguous 64-byte
// pieces, and the sums over the dimensions are 8-lane DPP reductions.  One wavefront per label, eight points at a time.
// ------------------------------------------------------------------------------------------------
struct ReduceMv8Args {
    int N, L, Nsrc, D;
    const double *tfeat;    // V x D vertex-major
    const TriRec *rec;
    const double *sfeat_vm; // Nsrc x D vertex-major copy of the moving features
    const double *cfw_vm;   // Nsrc x cfw_rows vertex-major copy of the weights, or nullptr
    int cfw_rows;
    const int *pptr, *pidx;
    const int *order;
    const double *absw;
    const int *stri;
    const double *sw3;
    int simmeasure;
    double *U;
};

__global__ __launch_bounds__(256) void k_unary_reduce_mv8(ReduceMv8Args a) {
    const int node = a.order[blockIdx.x];
    const int beg = a.pptr[node], P = a.pptr[node + 1] - beg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane / kMvLanes, j = lane % kMvLanes;
    const double absw = a.absw[node];
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const int D = a.D;
    for (int l = wave; l < a.L; l += 4) {
        const size_t g0 = (size_t)a.L * beg + (size_t)l * P;
        double acc = 0.0;  // this group's points (lane 0 of the group holds the value)
        bool bad = false;
        for (int i0 = 0; i0 < P; i0 += 64 / kMvLanes) {  // wavefront-uniform: the group sums need all lanes
            const int i = i0 + grp;
            const bool have = i < P;
            int t = -1, sv = 0;
            double wa = 0, wb = 0, wc = 0;
            if (have) {
                t = a.stri[g0 + i];
                sv = a.pidx[beg + i];
                wa = a.sw3[3 * (g0 + i)], wb = a.sw3[3 * (g0 + i) + 1], wc = a.sw3[3 * (g0 + i) + 2];
            }
            if (have && t < 0) bad = true;
            const bool go = have && t >= 0;
            const double *f0 = a.tfeat, *f1 = a.tfeat, *f2 = a.tfeat, *sa = a.sfeat_vm, *cw = nullptr;
            if (go) {
                const TriRec &r = a.rec[t];
                f0 = a.tfeat + (size_t)r.id[0] * D, f1 = a.tfeat + (size_t)r.id[1] * D, f2 = a.tfeat + (size_t)r.id[2] * D;
                sa = a.sfeat_vm + (size_t)sv * D;
                cw = a.cfw_vm ? a.cfw_vm + (size_t)sv * a.cfw_rows : nullptr;
            }
            const double c = feature_vector_similarity8(a.simmeasure, go, j, D, sa, cw, a.cfw_rows, f0, f1, f2, wa, wb, wc);
            if (go && j == 0) acc += c;
        }
        // the groups' partial sums: one value per group in its lane 0
        double tot = (j == 0) ? acc : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
        double cost = P > 0 ? tot / P : tot;
        if (__ballot(bad)) cost = nan;
        if (lane == 0) a.U[(size_t)l * a.N + node] = absw * cost;
    }
}

// ------------------------------------------------------------------------------------------------
// Patchwise reduction, D <= 8 * K, SSD / correlation (M/DiscreteCostFunction.cpp:680-692: per feature channel a patch
// similarity, averaged over the channels), in the layout of k_unary_reduce_mv8: eight lanes per patch point, lane j owns
// the channels j, j + 8, ...; here the sums run over the POINTS, so every lane accumulates its channels over its group's
// points and the eight groups of the wavefront are combined with three shuffle steps.  Two passes over the patch
// (means, then moments); the second pass re-reads the rows (cache hits) instead of keeping them in registers.
// ------------------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void k_unary_reduce_pw8(ReduceMv8Args a) {
    const int node = a.order[blockIdx.x];
    const int beg = a.pptr[node], P = a.pptr[node + 1] - beg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = lane / kMvLanes, j = lane % kMvLanes;
    const double absw = a.absw[node];
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const int D = a.D;
    auto across_groups = [](double v) {  // lanes with the same j: the eight groups of the wavefront
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        return v;
    };
    for (int l = wave; l < a.L; l += 4) {
        const size_t g0 = (size_t)a.L * beg + (size_t)l * P;
        bool bad = false;
        // one point of the patch for this lane's group: moving row, interpolated target row (this lane's channels), weight
        auto fetch = [&](int i, double *A, double *B, double &w) {
            const int t = a.stri[g0 + i];
            const int sv = a.pidx[beg + i];
            w = (a.cfw_vm && a.cfw_rows >= 1) ? a.cfw_vm[(size_t)sv * a.cfw_rows] : 1.0;
            if (t < 0) {
                bad = true;
#pragma unroll
                for (int k = 0; k < K; ++k) A[k] = B[k] = nan;
                return;
            }
            const TriRec &r = a.rec[t];
            const double *f0 = a.tfeat + (size_t)r.id[0] * D, *f1 = a.tfeat + (size_t)r.id[1] * D, *f2 = a.tfeat + (size_t)r.id[2] * D;
            const double *sa = a.sfeat_vm + (size_t)sv * D;
            const double wa = a.sw3[3 * (g0 + i)], wb = a.sw3[3 * (g0 + i) + 1], wc = a.sw3[3 * (g0 + i) + 2];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int d = j + kMvLanes * k;
                A[k] = d < D ? sa[d] : 0.0;
                B[k] = d < D ? wa * f0[d] + wb * f1[d] + wc * f2[d] : 0.0;
            }
        };
        double c[K];
        if (a.simmeasure == 2) {  // sparsesimkernel::corr over the patch points, per channel
            double sw = 0.0, ma[K], mb[K];
#pragma unroll
            for (int k = 0; k < K; ++k) ma[k] = mb[k] = 0.0;
            for (int i = grp; i < P; i += 64 / kMvLanes) {
                double A[K], B[K], w;
                fetch(i, A, B, w);
                sw += w;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    ma[k] += w * A[k];
                    mb[k] += w * B[k];
                }
            }
            sw = across_groups(sw);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                ma[k] = across_groups(ma[k]);
                mb[k] = across_groups(mb[k]);
                if (sw > 0.0) {
                    ma[k] /= sw;
                    mb[k] /= sw;
                }
            }
            double pr[K], va[K], vb[K];
#pragma unroll
            for (int k = 0; k < K; ++k) pr[k] = va[k] = vb[k] = 0.0;
            for (int i = grp; i < P; i += 64 / kMvLanes) {
                double A[K], B[K], w;
                fetch(i, A, B, w);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const double da = A[k] - ma[k], db = B[k] - mb[k];
                    pr[k] += w * da * db;
                    va[k] += w * da * da;
                    vb[k] += w * db * db;
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                double p = across_groups(pr[k]), x = across_groups(va[k]), y = across_groups(vb[k]);
                if (sw > 0.0) {
                    p /= sw;
                    x /= sw;
                    y /= sw;
                }
                const double rr = (x == 0.0 || y == 0.0) ? 0.0 : p / (sqrt(x) * sqrt(y));
                c[k] = 1 - (1 + rr) * 0.5;
            }
        } else {  // sparsesimkernel::SSD
            double pr[K];
#pragma unroll
            for (int k = 0; k < K; ++k) pr[k] = 0.0;
            for (int i = grp; i < P; i += 64 / kMvLanes) {
                double A[K], B[K], w;
                fetch(i, A, B, w);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const double df = A[k] - B[k];
                    pr[k] += w * df * df;
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) c[k] = sqrt(across_groups(pr[k])) / P;
        }
        double total = 0.0;  // this lane's channels, then the eight lanes of the group (all groups hold the same values)
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (j + kMvLanes * k < D) total += c[k];
        total = mv_group_sum(total);
        double cost = total / D;
        if (__ballot(bad)) cost = nan;
        if (lane == 0) a.U[(size_t)l * a.N + node] = absw * cost;
    }
}

// ------------------------------------------------------------------------------------------------
// launch
// ------------------------------------------------------------------------------------------------
// dynamic LDS of k_unary_samples when the labels are split over nsplit workgroups
static size_t samples_lds(int pmax, int L, int nsplit) {
    const size_t lper = (size_t)(L + nsplit - 1) / nsplit;
    return sizeof(double) * (5 * (size_t)pmax + 9 * (size_t)L + lper * pmax) + sizeof(unsigned) * kQueueCap + sizeof(int) * 2 * kChunk;
}

// Workgroups per control point: enough that one workgroup's samples are about one LDS pass (at most 4 for that
// reason), and more when the patch is so large (coarse control grid under a fine data grid) that the per-label target
// values would not fit in LDS otherwise.
int unary_nsplit(int L, int pmax) {
    if (const char *e = std::getenv("MSMHIP_NSPLIT")) return std::max(1, std::min(L, atoi(e)));  // experiments only
    int n = std::max(1, std::min(4, (int)(((size_t)L * pmax + kChunk - 1) / kChunk)));
    while (n < L && samples_lds(pmax, L, n) > 64 * 1024) ++n;
    return std::min(n, std::max(L, 1));
}
int unary_fix_segments() { return kFixSegs; }
size_t unary_fix_counter_words() { return (size_t)kCntStride * (1 + kFixSegs); }
// samples of workgroup `block` of the unary sample kernels (the mapping at the top of k_unary_samples / k_unary_rays)
void unary_fix_offsets(int N, int L, int pmax, const int32_t *pptr, const int32_t *order, std::vector<uint32_t> &off) {
    const int nsplit = unary_nsplit(L, pmax), per = (N + 7) >> 3, slots = 8 * per;
    const int lper = (L + nsplit - 1) / nsplit;
    std::vector<uint64_t> size(kFixSegs, 0);
    for (int b = 0; b < nsplit * slots; ++b) {
        const int part = b / slots, slot = b - part * slots;
        const int at = (slot & 7) * per + (slot >> 3);
        if (at >= N) continue;
        const int node = order[at];
        const int l_beg = part * lper, l_end = std::min(L, l_beg + lper);
        if (l_beg >= l_end) continue;
        size[b % kFixSegs] += (uint64_t)(l_end - l_beg) * (pptr[node + 1] - pptr[node]);
    }
    off.assign(kFixSegs + 1, 0);
    for (int s = 0; s < kFixSegs; ++s) off[s + 1] = (uint32_t)std::min<uint64_t>(0xffffffffull, off[s] + size[s]);
}

static bool uses_ray_table(const DevTree &t) { return t.simple && t.ray_G > 0 && t.ray_cell && t.ray_tri; }

static int launch_samples(msm_ctx *ctx, const UnaryLaunch &u, const UnaryWeightsScratch *w, SamplesArgs &a, bool fused_reduction = false) {
    if (u.tree.nnodes <= 0 || !u.tree.mask) return fail(MSM_ERR_STATE, "target search structure missing");
    a.tree = u.tree;
    a.tfeat = u.tfeat;
    a.tD = u.D > 0 ? u.D : 1;
    a.N = u.N;
    a.L = u.L;
    a.rnl = u.rnl;
    a.src = u.src;
    a.Nsrc = u.Nsrc;
    a.pptr = u.pptr;
    a.pidx = u.pidx;
    a.order = u.order;
    a.pmax = u.pmax;
    a.tval = w ? nullptr : u.tval;
    a.stri = w ? w->stri : nullptr;
    a.sw3 = w ? w->sw3 : nullptr;
    a.fix_list = u.fix_list;
    a.fix_pt = u.fix_pt;
    a.fix_cnt = u.fix_cnt;
    a.fix_off = u.fix_off;
    a.sfeat = u.sfeat;
    a.cfw = (u.cfw && u.cfw_rows >= 1) ? u.cfw : nullptr;
    a.absw = u.absw;
    a.simmeasure = u.simmeasure;
    a.percentile = u.percentile;
    a.U = (w || !fused_reduction) ? nullptr : u.U;  // the fused reduction is the univariate one
    a.redo_list = u.redo_list;
    a.redo_count = u.fix_cnt;
    a.status = ctx->d_status;
    if (u.ntri >= (1 << kTriBits)) return fail(MSM_ERR_CAPACITY, "target mesh has %d triangles; the sample queue packs ids in %d bits", u.ntri, kTriBits);
    a.nsplit = unary_nsplit(u.L, u.pmax);
    const size_t lds = samples_lds(u.pmax, u.L, a.nsplit);
    if (!uses_ray_table(u.tree) && lds > 64 * 1024) {
        if (lds > 160 * 1024) return fail(MSM_ERR_CAPACITY, "a patch of %d points does not fit in LDS", u.pmax);
        MSM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_unary_samples), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const int blocks = a.nsplit * 8 * ((u.N + 7) / 8);
    if (u.ev_start) MSM_HIP(hipEventRecord(u.ev_start, ctx->stream));
    if (uses_ray_table(u.tree)) {
        a.U = nullptr;  // the reduction is a kernel of its own on this path
        const int lper = (u.L + a.nsplit - 1) / a.nsplit;
        const size_t rays_lds = sizeof(double) * (3 * (size_t)u.pmax + 9 * (size_t)u.L) + sizeof(int) * ((size_t)lper * u.pmax + 4) + 16;
        if (rays_lds > 64 * 1024) MSM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_unary_rays), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rays_lds));
        hipLaunchKernelGGL(k_unary_rays, dim3(blocks), dim3(256), rays_lds, ctx->stream, a);
    } else {
        hipLaunchKernelGGL(k_unary_samples, dim3(blocks), dim3(256), lds, ctx->stream, a);
    }
    MSM_HIP(hipGetLastError());
    if (u.ev_stop) MSM_HIP(hipEventRecord(u.ev_stop, ctx->stream));
    hipLaunchKernelGGL(k_unary_fixup, dim3(4096), dim3(256), 0, ctx->stream, a);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_unary_univariate(msm_ctx *ctx, const UnaryLaunch &u) {
    SamplesArgs a;
    const bool dice = u.simmeasure == 4 || u.simmeasure == 5;
    // Correlation / SSD: both search paths only fill tval and one kernel reduces it, so the table does not depend on
    // which of them ran (the ray table of a target arrives in the background, api.cpp: ensure_rays).  DICE costs are
    // ratios of counts -- the same from any summation order -- and keep the reduction fused into the general kernel.
    int st = launch_samples(ctx, u, nullptr, a, dice);
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
    r.percentile = u.percentile;
    r.U = u.U;
    r.redo_list = u.redo_list;
    r.redo_count = u.fix_cnt;
    if (uses_ray_table(u.tree) && dice) {
        // the rank-counting DICE reduction is the wavefront-per-label kernel; all control points
        r.redo_list = nullptr;
        r.redo_count = nullptr;
        hipLaunchKernelGGL(k_unary_reduce_univariate, dim3(std::min(u.N, 2048)), dim3(256), sizeof(double) * 2 * (size_t)u.pmax, ctx->stream, r);
        MSM_HIP(hipMemsetAsync(u.fix_cnt, 0, unary_fix_counter_words() * sizeof(unsigned), ctx->stream));
    } else if (!dice) {
        r.redo_list = u.order;  // all control points, in launch order
        hipLaunchKernelGGL(k_unary_reduce_flat, dim3(u.N), dim3(256), sizeof(double) * 2 * (size_t)u.pmax, ctx->stream, r, u.fix_cnt,
                           (int)unary_fix_counter_words());
    } else {
        hipLaunchKernelGGL(k_unary_reduce_univariate, dim3(64), dim3(256), sizeof(double) * 2 * (size_t)u.pmax, ctx->stream, r);
        MSM_HIP(hipMemsetAsync(u.fix_cnt, 0, unary_fix_counter_words() * sizeof(unsigned), ctx->stream));  // counters are zero between launches
    }
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
    r.pmax = u.pmax;
    r.percentile = u.percentile;
    r.U = u.U;
    const bool dice = u.simmeasure == 4 || u.simmeasure == 5;
    if (!dice && u.D >= 12 && u.D <= kMvLanes * kMvKeep && u.sfeat_vm) {  // few dimensions: a lane per point wastes less
        ReduceMv8Args m;
        m.N = u.N, m.L = u.L, m.Nsrc = u.Nsrc, m.D = u.D;
        m.tfeat = u.tfeat;
        m.rec = u.tree.rec;
        m.sfeat_vm = u.sfeat_vm;
        m.cfw_vm = u.cfw_vm;
        m.cfw_rows = u.cfw_rows;
        m.pptr = u.pptr, m.pidx = u.pidx, m.order = u.order;
        m.absw = u.absw;
        m.stri = w.stri, m.sw3 = w.sw3;
        m.simmeasure = u.simmeasure;
        m.U = u.U;
        if (!patchwise) hipLaunchKernelGGL(k_unary_reduce_mv8, dim3(u.N), dim3(256), 0, ctx->stream, m);
        else if (u.D <= 32) hipLaunchKernelGGL(k_unary_reduce_pw8<4>, dim3(u.N), dim3(256), 0, ctx->stream, m);
        else hipLaunchKernelGGL(k_unary_reduce_pw8<8>, dim3(u.N), dim3(256), 0, ctx->stream, m);
    } else {
        size_t flds = (dice && patchwise) ? sizeof(double) * 8 * (size_t)u.pmax : 0;
        int threads = 256;
        if (flds > 160 * 1024) {  // one wavefront per control point: a quarter of the LDS
            flds /= 4;
            threads = 64;
        }
        if (flds > 160 * 1024) return fail(MSM_ERR_CAPACITY, "a patch of %d points does not fit in LDS (patchwise DICE)", u.pmax);
        if (flds > 64 * 1024) MSM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_unary_reduce_features), hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds));
        hipLaunchKernelGGL(k_unary_reduce_features, dim3(u.N), dim3(threads), flds, ctx->stream, r);
    }
    MSM_HIP(hipGetLastError());
    MSM_HIP(hipMemsetAsync(u.fix_cnt, 0, unary_fix_counter_words() * sizeof(unsigned), ctx->stream));  // counters are zero between launches
    return MSM_OK;
}

}  // namespace msm
