// move_kernels.hip -- one label step of Fusion::optimize for the triclique (HO) cost classes (I/Fusion/Fusion.h:181-196:
// eight computeTripletCost calls per control triangle, M/DiscreteCostFunction.cpp:135-188 with HO*::triplet_likelihood
// :487-531 / :565-618) on a direction-table target, as ONE kernel.
//
// What every HCP configuration evaluates 38 times per iteration.  Per move: 8 x T evaluations, each sampling the ~8 source
// vertices binned under its control triangle: 327 696 samples at ico6 / ico4 -- too few to be a throughput problem on 256
// CUs; the move is a chain of dependent steps (labels -> proposed triangle -> sample position -> direction cell -> triangle
// record -> value -> similarity -> cost), and its time is the length of that chain times the number of rounds the
// workgroups need to get onto the chip.  The three-kernel version of round 1 (sample / fix up / reduce: 31 + 15 + 16 us,
// plus 58 us of copy commands and status read-backs per call) is replaced by:
//   * k_move_prepare, once per get_source_data(): everything that only depends on the CURRENT control grid -- the source
//     vertex projected on its control triangle and its barycentric coordinates there (:498-505 / :574-583), the moving
//     patch's weighted mean / variance (M/similarities.cpp:135-150), the original triangle's half of the strain energy
//     (M/reg_tools.cpp:698-715) -- same arithmetic, same bits, just not 38 x 8 times per iteration;
//   * k_ho_move, one launch per move.  A workgroup takes a run of <= 8 control triangles (<= 64 bin slots, host packed:
//     640 workgroups at ico4, all resident at once -- a second round of workgroups would double the chain).  A lane per
//     (triangle, combination) sets up the proposed triangles in LDS (fold test included); then every lane samples
//     (combination, bin point) pairs, neighbouring lanes taking neighbouring bin points; then the lane per evaluation
//     reduces it from LDS (similarity in the reference's serial order + strain) and writes the cost.  No value goes
//     through HBM;
//   * samples the direction table cannot settle (0.2 %) are listed in LDS and searched in the octree leaf by eight lanes
//     each (search_device.hpp: group8_find) before the reduction; what even that cannot decide (no candidate in the leaf: sibling
//     leaves, nearest vertex -- a handful per registration) marks the evaluation for the tail kernel, which the HOST
//     launches only when the main kernel raised the flag;
//   * the labeling arrives in the kernel arguments (a byte per control point) and the costs are written straight into
//     mapped pinned host memory, so a call is one launch and one synchronisation.
#include "clique_device.hpp"

namespace msm {

namespace {

constexpr unsigned long long kPendingBits = 0x7ff8dead0badc0deull;  // a NaN no computation produces: "this sample is left to the tail"
__device__ __forceinline__ double pending_value() { return __longlong_as_double((long long)kPendingBits); }
__device__ __forceinline__ bool is_pending(double v) { return (unsigned long long)__double_as_longlong(v) == kPendingBits; }

template <bool kPacked>
__device__ __forceinline__ int label_of(const MoveArgs &m, const MoveLabels &lab, int node) {
    if (kPacked) return (int)((lab.w[node >> 2] >> ((node & 3) * 8)) & 255u);
    return m.labeling[node];
}

__device__ __forceinline__ void raise_both(const CliqueArgs &a, const MoveArgs &m, int code) {
    atomicMin(a.status, code);
    __hip_atomic_store(m.host_flags, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// s / P for 0 <= s < 2^22, P > 0, with a float reciprocal and an exact correction step
__device__ __forceinline__ int fast_div(int s, int P, float invP) {
    int q = (int)((float)s * invP);
    if (q * P > s) --q;
    else if ((q + 1) * P <= s) ++q;
    return q;
}

// position of a bin point for a proposed triangle (n0, n1, n2 = g[0..9)): HO*::get_target_data :506-510 / :584-590 from
// the point's barycentric coordinates in the current control triangle
__device__ __forceinline__ V3 moved_point(const double *g, double wa, double wb, double wc) {
    const V3 tmp = mk(g[0] * wa + g[3] * wb + g[6] * wc, g[1] * wa + g[4] * wb + g[7] * wc, g[2] * wa + g[5] * wb + g[8] * wc);
    return scale(normalized(tmp), kRad);
}

// Direction-table search that also hands back the winning triangle's vertices (and single feature) from its table
// record: the first candidate's vertices are requested together with its edge planes, as in k_unary_rays.
__device__ __forceinline__ int ray_find_rec(const DevTree &T, const V3 &p, double2 &d0, double2 &d1, double2 &d2, double2 &d3, double2 &d4, double2 &d5) {
    float fx, fy, fz;
    const int4 c = ray_cell_of(T, p, fx, fy, fz);
    if (c.x < 0) return -1;
    const float4 *rec = T.ray_tri + (size_t)kRayPieces * c.x;
    const float4 e0 = rec[0], e1 = rec[1], e2 = rec[2];
    const double2 *dv = reinterpret_cast<const double2 *>(rec + 3);
    d0 = dv[0], d1 = dv[1], d2 = dv[2], d3 = dv[3], d4 = dv[4], d5 = dv[5];
    int t = c.x;
    float4 ev = e1;
    float least_a = 0.f, least;
    const int lvl0 = ray_accept_level(e0, e1, e2, fx, fy, fz, least_a);
    if (lvl0 != 2) {
        t = -1;
        int near_a = lvl0 == 1 ? c.x : -1, near_b = -1;  // the candidates the float test nearly accepted, likeliest first (see below)
        int4 mo = make_int4(c.w, -1, -1, -1);
        if (c.w < -1) mo = T.ray_more[-2 - c.w];
#pragma unroll 1
        for (int k = 0; k < 6 && t < 0; ++k) {
            const int ck = k == 0 ? c.y : (k == 1 ? c.z : (k == 2 ? mo.x : (k == 3 ? mo.y : (k == 4 ? mo.z : mo.w))));
            if (ck < 0) break;
            const float4 *r2 = T.ray_tri + (size_t)kRayPieces * ck;
            const float4 g0 = r2[0], g1 = r2[1], g2 = r2[2];
            const int lvl = ray_accept_level(g0, g1, g2, fx, fy, fz, least);
            if (lvl == 2) {
                t = ck;
                ev = g1;
            } else if (lvl == 1) {
                if (near_a < 0 || least > least_a) {
                    near_b = near_a;
                    near_a = ck;
                    least_a = least;
                } else if (near_b < 0) {
                    near_b = ck;
                }
            }
        }
        if (t >= 0) {
            dv = reinterpret_cast<const double2 *>(T.ray_tri + (size_t)kRayPieces * t + 3);
            d0 = dv[0], d1 = dv[1], d2 = dv[2], d3 = dv[3], d4 = dv[4], d5 = dv[5];
        }
#ifndef MSM_MOVE_NO_FP64_RETEST  // (diagnostics: the kernel without this step)
        else if (near_a >= 0) {
            // No candidate passed in float: 3e-6 of the stored threshold are the float evaluation's allowance, thirty times the margin the
            // proof needs (octree.cpp: build_ray_table).  The (at most two) candidates the float test nearly accepted are looked at again
            // with FP64 edge planes from the record's vertices, which leaves open one sample in 15 000 instead of one in 1 000 -- and with
            // them the leaf search of the open samples that every third workgroup had to run.
            const double pn = norm(p);
#pragma unroll 1
            for (int r = 0; r < 2 && t < 0; ++r) {
                const int ck = r == 0 ? near_a : near_b;
                if (ck < 0) break;
                if (ck == c.x) {  // the first candidate's record is here already
                    if (ray_accepts_fp64(mk(d0.x, d0.y, d1.x), mk(d1.y, d2.x, d2.y), mk(d3.x, d3.y, d4.x), p, pn, (double)e0.w - kRayFloatAllowance + 1e-12)) {
                        t = ck;
                        ev = e1;
                    }
                } else {
                    const float4 *r2 = T.ray_tri + (size_t)kRayPieces * ck;
                    const float thr = r2[0].w;
                    const float4 g1 = r2[1];
                    const double2 *dq = reinterpret_cast<const double2 *>(r2 + 3);
                    const double2 q0 = dq[0], q1 = dq[1], q2 = dq[2], q3 = dq[3], q4 = dq[4];
                    if (ray_accepts_fp64(mk(q0.x, q0.y, q1.x), mk(q1.y, q2.x, q2.y), mk(q3.x, q3.y, q4.x), p, pn, (double)thr - kRayFloatAllowance + 1e-12)) {
                        t = ck;
                        ev = g1;
                        d0 = q0, d1 = q1, d2 = q2, d3 = q3, d4 = q4, d5 = dq[5];
                    }
                }
            }
        }
#endif
    }
    if (t >= 0 && __float_as_int(ev.w) >= 0 && !ray_vouches(T, ev, p)) t = -1;  // its leaf may not list the triangle
    return t;
}
#ifdef MSM_MOVE_TRACE
// diagnostics: why the table left a sample open -- 0 no cell, 1 no candidate near its threshold, 2 near but outside in FP64, 3 its leaf may not list it
__device__ int ray_open_reason(const DevTree &T, const V3 &p) {
    float fx, fy, fz;
    const int4 c = ray_cell_of(T, p, fx, fy, fz);
    if (c.x < 0) return 0;
    int4 mo = make_int4(c.w, -1, -1, -1);
    if (c.w < -1) mo = T.ray_more[-2 - c.w];
    const double pn = norm(p);
    int reason = 1;
    for (int k = 0; k < 7; ++k) {
        const int ck = k == 0 ? c.x : (k == 1 ? c.y : (k == 2 ? c.z : (k == 3 ? mo.x : (k == 4 ? mo.y : (k == 5 ? mo.z : mo.w)))));
        if (ck < 0) break;
        const float4 *r2 = T.ray_tri + (size_t)kRayPieces * ck;
        const float4 g0 = r2[0], g1 = r2[1], g2 = r2[2];
        float least;
        const int lvl = ray_accept_level(g0, g1, g2, fx, fy, fz, least);
        if (lvl == 0) continue;
        reason = 2;
        const double2 *dq = reinterpret_cast<const double2 *>(r2 + 3);
        const double2 q0 = dq[0], q1 = dq[1], q2 = dq[2], q3 = dq[3], q4 = dq[4];
        if (lvl == 2 || ray_accepts_fp64(mk(q0.x, q0.y, q1.x), mk(q1.y, q2.x, q2.y), mk(q3.x, q3.y, q4.x), p, pn, (double)g0.w - kRayFloatAllowance + 1e-12)) return 3;
    }
    return reason;
}
#endif

// the eight proposed triangles of control triangle t: combination k (bits A,B,C; 0 = current label), I/Fusion/Fusion.h:188-195
template <bool kPacked>
__device__ __forceinline__ bool proposed_triangle(const CliqueArgs &a, const MoveArgs &m, const MoveLabels &lab, int t, int k, int *id, V3 *r) {
    id[0] = a.triplets[3 * t], id[1] = a.triplets[3 * t + 1], id[2] = a.triplets[3 * t + 2];
    const int la = (k & 4) ? m.label : label_of<kPacked>(m, lab, id[0]);
    const int lb = (k & 2) ? m.label : label_of<kPacked>(m, lab, id[1]);
    const int lc = (k & 1) ? m.label : label_of<kPacked>(m, lab, id[2]);
    r[0] = aos(a.moved, (size_t)id[0] * a.L + la), r[1] = aos(a.moved, (size_t)id[1] * a.L + lb), r[2] = aos(a.moved, (size_t)id[2] * a.L + lc);
    const V3 c0 = soa(a.cp, a.N, id[0]), c1 = soa(a.cp, a.N, id[1]), c2 = soa(a.cp, a.N, id[2]);
    return dot(tri_normal(r[0], r[1], r[2]), tri_normal(c0, c1, c2)) < 0.0;  // computeTripletCost :151-152
}

// HO*::triplet_likelihood of one evaluation from its point values vals[0..n), in the reference's serial operand order;
// the moving patch's half of the correlation (sum of weights, weighted mean, weighted variance, M/similarities.cpp:135-150)
// comes from k_move_prepare: st[0..3), and W * (A - meanA) per slot
__device__ __forceinline__ double move_likelihood(const CliqueArgs &a, int n, const double *st, const double *wda, const double *cw, const double *sf, double wmean,
                                                  const double *vals) {
    for (int i = 0; i < n; ++i)
        if (vals[i] != vals[i]) return __longlong_as_double(0x7ff8000000000000ll);  // a failed search
    double cost = 0.0;
    if (a.kind == MSM_COST_HO_UNIVARIATE) {
        if (a.simmeasure == 2) {
            const double sum = st[0], varA = st[2];
            double meanB = 0.0, prod = 0.0, varB = 0.0;
            for (int i = 0; i < n; ++i) meanB += (cw ? cw[i] : 1.0) * vals[i];
            if (sum > 0.0) meanB /= sum;
            for (int i = 0; i < n; ++i) {
                prod += wda[i] * (vals[i] - meanB);
                varB += (cw ? cw[i] : 1.0) * (vals[i] - meanB) * (vals[i] - meanB);
            }
            if (sum > 0.0) {
                prod /= sum;
                varB /= sum;
            }
            const double r = (varA == 0.0 || varB == 0.0) ? 0.0 : prod / (sqrt(varA) * sqrt(varB));
            cost = 1 - (1 + r) * 0.5;
        } else if (a.simmeasure == 4 || a.simmeasure == 5) {  // sparsesimkernel::DICE / genDICE, :201-253 (weights unused)
            cost = dice_serial(a.simmeasure, n, a.percentile, [&](int i) { return sf[i]; }, [&](int i) { return vals[i]; });
        } else {  // sparsesimkernel::SSD, :179-188
            double prod = 0.0;
            for (int i = 0; i < n; ++i) prod += (cw ? cw[i] : 1.0) * (sf[i] - vals[i]) * (sf[i] - vals[i]);
            cost = sqrt(prod) / n;
        }
    } else {
        for (int i = 0; i < n; ++i) cost += vals[i];
        if (n > 0) cost /= n;
    }
    return wmean * cost;
}

__device__ __forceinline__ double move_strain(const CliqueArgs &a, const double *fp, const V3 *r) {
    StrainFrame fr;
    fr.i00 = fp[0], fr.i01 = fp[1], fr.i10 = fp[2], fr.i11 = fp[3];
    fr.dswap = fp[4] != 0.0;
    return a.lambda * pow_exp(triangular_strain_from(fr, r, a.mu, a.kappa, a.k_exp), a.rexp);
}

}  // namespace

// per control triangle: the label-independent half of HO*::get_target_data for each of its bin points, the moving patch's
// half of the correlation, and the original triangle's half of the strain energy
__global__ __launch_bounds__(256) void k_move_prepare(CliqueArgs a, int *__restrict__ slot_tri, double *__restrict__ slot_w, double *__restrict__ slot_sf,
                                                       double *__restrict__ slot_cw, double *__restrict__ slot_wda, double *__restrict__ tri_frame,
                                                       double *__restrict__ tri_stat) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.T) return;
    const int id[3] = {a.triplets[3 * t], a.triplets[3 * t + 1], a.triplets[3 * t + 2]};
    const V3 org[3] = {soa(a.orig, a.Norig, id[0]), soa(a.orig, a.Norig, id[1]), soa(a.orig, a.Norig, id[2])};
    const StrainFrame fr = strain_frame(org);
    double *fp = tri_frame + 5 * (size_t)t;
    fp[0] = fr.i00, fp[1] = fr.i01, fp[2] = fr.i10, fp[3] = fr.i11, fp[4] = fr.dswap ? 1.0 : 0.0;
    const V3 cp0 = soa(a.cp, a.N, id[0]), cp1 = soa(a.cp, a.N, id[1]), cp2 = soa(a.cp, a.N, id[2]);
    V3 s3;
    double pd;
    plane_of(cp0, cp1, cp2, s3, pd);
    const int beg = a.bin_ptr[t], end = a.bin_ptr[t + 1];
    for (int s = beg; s < end; ++s) {
        const int sv = a.bin_idx[s];
        const V3 sp = project_with_plane(soa(a.src, a.Nsrc, sv), s3, pd);  // project_point(src, cp0, cp1, cp2)
        double wa, wb, wc;
        area_weights(cp0, cp1, cp2, sp, wa, wb, wc);  // barycentric(), R/triangle.cpp:159-172
        slot_tri[s] = t;
        slot_w[3 * (size_t)s] = wa, slot_w[3 * (size_t)s + 1] = wb, slot_w[3 * (size_t)s + 2] = wc;
        slot_sf[s] = a.sfeat[sv];                 // feature row 1 (the univariate classes; unused by the multivariate one)
        if (slot_cw) slot_cw[s] = a.cfw[sv];      // weight row 1
    }
    // sparsesimkernel::corr, M/similarities.cpp:135-150, the sums that involve the moving patch only (serial, as there)
    auto W = [&](int s) { return a.cfw ? a.cfw[a.bin_idx[s]] : 1.0; };
    auto A = [&](int s) { return a.sfeat[a.bin_idx[s]]; };
    double sum = 0.0, meanA = 0.0, varA = 0.0;
    for (int s = beg; s < end; ++s) sum += W(s);
    for (int s = beg; s < end; ++s) meanA += W(s) * A(s);
    if (sum > 0.0) meanA /= sum;
    for (int s = beg; s < end; ++s) {
        varA += W(s) * (A(s) - meanA) * (A(s) - meanA);
        slot_wda[s] = W(s) * (A(s) - meanA);
    }
    if (sum > 0.0) varA /= sum;
    tri_stat[3 * (size_t)t] = sum, tri_stat[3 * (size_t)t + 1] = meanA, tri_stat[3 * (size_t)t + 2] = varA;
}

// kMode 0: HO univariate (value = interpolated target feature, from the table record); 3: as 2 with two dimension pairs per lane (D <= 32)
//       1: HO multivariate, a lane per sample (any D, any measure): ho_value_on
//       2: HO multivariate, 12 <= D <= 64 and even, SSD / correlation: eight lanes per sample split the dimensions
template <bool kPacked, int kMode, int kThreads>
__global__ __launch_bounds__(kThreads) void k_ho_move(CliqueArgs a, MoveArgs m, MoveLabels lab) {
    extern __shared__ __align__(16) double lds[];
    double *s_geo = lds;                            // 64 evaluations x 9: the proposed triangles
    double *s_vals = s_geo + 64 * 9;                // 8 combinations x cap bin slots
    // what the last phase needs besides the samples -- label independent, fetched at the very start by the lanes that have nothing
    // to do while lanes 0 .. 8 * ntrip set up the proposed triangles, so that the last phase starts without dependent global loads
    double *s_wda = s_vals + 8 * (size_t)m.cap, *s_cw = s_wda + m.cap, *s_sf = s_cw + m.cap;  // per bin slot
    double *s_stat = s_sf + m.cap, *s_frame = s_stat + 24, *s_strain = s_frame + 40;            // 8 x 3, 8 x 5, 64
    double *s_w = s_strain + 64;                     // kMode 2: 3 weights per sample of a round, then its similarity's moments (4)
    int *s_flag = reinterpret_cast<int *>(s_w + (kMode >= 2 ? 4 * kThreads : 0));  // [0,64) folded, [64,128) deferred
    int *s_pend = s_flag + 128;                     // samples the direction table left open (at most all of a round... of the block: 8 * cap)
    int *s_tt = s_pend + 8 * m.cap;                 // kMode 2: triangle and its vertex ids per sample of a round (4 x kThreads)
    int *s_bin = s_tt + (kMode >= 2 ? 4 * kThreads : 0);  // bin_ptr[t0 .. t0 + ntrip]
    __shared__ int s_npend;

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef MSM_MOVE_TRACE  // diagnostics: phase time stamps of every workgroup (tools/move_trace.py)
#define MSM_STAMP(k) do { if (m.trace && tid == 0) m.trace[8 * (size_t)blockIdx.x + (k)] = wall_clock64(); } while (0)
#else
#define MSM_STAMP(k) do { } while (0)
#endif
    MSM_STAMP(0);
    if (blockIdx.x == 0 && tid == 0) m.defer_cnt[m.parity ^ 1] = 0u;  // the previous move's list has been consumed
    const int per = (m.nblk + 7) >> 3;  // the workgroups of an XCD (blockIdx % 8) take a contiguous run of control triangles
    const int blk = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (blk >= m.nblk) return;
    const int4 bk = m.blk[blk];
    const int t0 = bk.x, ntrip = bk.y, s0 = bk.z, nslots = bk.w;
    const int total = (m.single ? 1 : 8) * nslots;  // single: combination 000 only (evaluateTotalCostSum's triplet part)
    const float inv = 1.0f / (float)max(nslots, 1);

    // ---- the eight proposed triangles of each control triangle (lanes 0 .. 8 * ntrip); the other lanes meanwhile fetch their
    // first sample's slot data, which does not depend on the labels
    const bool ev = tid < 8 * ntrip && !(m.single && (tid & 7));
    if (tid == 0) s_npend = 0;
    if (tid >= kThreads - 64) {  // the last wavefront: the last phase's operands
        const int h = tid - (kThreads - 64);
        for (int j = h; j < nslots; j += 64) {
            s_wda[j] = m.slot_wda[s0 + j];
            s_sf[j] = m.slot_sf[s0 + j];
            if (m.slot_cw) s_cw[j] = m.slot_cw[s0 + j];
        }
        if (h < 3 * ntrip) s_stat[h] = m.tri_stat[3 * (size_t)t0 + h];
        if (h < 5 * ntrip) s_frame[h] = m.tri_frame[5 * (size_t)t0 + h];
        if (h <= ntrip) s_bin[h] = a.bin_ptr[t0 + h];
    }
    int id[3] = {0, 0, 0};
    double wmean = 0.0;  // mean AbsoluteWeight of the triangle's control points (HO*::triplet_likelihood :530 / :616)
    if (ev) {
        V3 r[3];
        const bool folded = proposed_triangle<kPacked>(a, m, lab, t0 + (tid >> 3), tid & 7, id, r);
        wmean = (a.absw[id[0]] + a.absw[id[1]] + a.absw[id[2]]) / 3.0;
        double *g = s_geo + 9 * tid;
        g[0] = r[0].x, g[1] = r[0].y, g[2] = r[0].z, g[3] = r[1].x, g[4] = r[1].y, g[5] = r[1].z, g[6] = r[2].x, g[7] = r[2].y, g[8] = r[2].z;
        s_flag[tid] = folded ? 1 : 0;
        s_flag[64 + tid] = 0;
    }
    MSM_STAMP(1);

    // ---- samples: s = combination * nslots + slot (neighbouring lanes = neighbouring bin points of one combination)
    for (int base = 0; base < total; base += kThreads) {
        const int s = base + tid;
        int tt = -2;  // -2: nothing to do (past the end, or a folded proposal: it never looks at the data); -1: left open
        int kk = 0, j = 0, el = 0;
        double wa = 0.0, wb = 0.0, wc = 0.0;
        if (s < total) {  // the slot data do not depend on the labels: requested before the first barrier
            kk = fast_div(s, nslots, inv), j = s - kk * nslots;
            const size_t slot = (size_t)(s0 + j);
            el = (m.slot_tri[slot] - t0) * 8 + kk;
            wa = m.slot_w[3 * slot], wb = m.slot_w[3 * slot + 1], wc = m.slot_w[3 * slot + 2];
        }
        if (base == 0) __syncthreads();  // the proposed triangles are in LDS
        if (base == 0) MSM_STAMP(2);
        if (s < total && !s_flag[el]) {
            const V3 p = moved_point(s_geo + 9 * el, wa, wb, wc);
            double2 d0, d1, d2, d3, d4, d5;
            tt = ray_find_rec(a.tree, p, d0, d1, d2, d3, d4, d5);
            if (tt >= 0) {
                area_weights(mk(d0.x, d0.y, d1.x), mk(d1.y, d2.x, d2.y), mk(d3.x, d3.y, d4.x), p, wa, wb, wc);
                if (kMode == 0) {
                    s_vals[kk * m.cap + j] = wa * d4.y + wb * d5.x + wc * d5.y;
                } else if (kMode == 1) {
                    s_vals[kk * m.cap + j] = ho_value_on(a, a.bin_idx[s0 + j], p, tt);
                } else {
                    const TriRec &rr = a.tree.rec[tt];
                    s_w[3 * tid] = wa, s_w[3 * tid + 1] = wb, s_w[3 * tid + 2] = wc;
                    s_tt[kThreads + 3 * tid] = rr.id[0], s_tt[kThreads + 3 * tid + 1] = rr.id[1], s_tt[kThreads + 3 * tid + 2] = rr.id[2];
                }
            } else {
                s_pend[atomicAdd(&s_npend, 1)] = s;
#ifdef MSM_MOVE_TRACE
                if (m.trace) atomicAdd(m.trace + 8 * (size_t)gridDim.x + ray_open_reason(a.tree, p), 1ull);
#endif
            }
        }
        if (kMode >= 2) {
            // eight lanes per sample: 32 groups x 8 passes cover the round's 256 samples (four lanes per sample and half the
            // passes took 112 instead of 95 us at D = 32: the passes are bound by the rows' way through L2, not by their number)
            s_tt[tid] = tt;
            __syncthreads();
            const int grp = tid >> 3, jj = tid & 7, D = a.D;
#pragma unroll 2
            for (int pass = 0; pass < 8; ++pass) {
                const int q = pass * (kThreads / 8) + grp, sq = base + q;
                const bool go = s_tt[q] >= 0;
                if (!__any(go)) continue;
                const double *f0 = a.tfeat, *f1 = a.tfeat, *f2 = a.tfeat, *sa = a.sfeat_vm, *cw = nullptr;
                int qk = 0, qj = 0;
                if (go) {
                    qk = fast_div(sq, nslots, inv), qj = sq - qk * nslots;
                    const int sv = a.bin_idx[s0 + qj];
                    f0 = a.tfeat + (size_t)s_tt[kThreads + 3 * q] * D, f1 = a.tfeat + (size_t)s_tt[kThreads + 3 * q + 1] * D, f2 = a.tfeat + (size_t)s_tt[kThreads + 3 * q + 2] * D;
                    sa = a.sfeat_vm + (size_t)sv * D;
                    cw = a.cfw_vm ? a.cfw_vm + (size_t)sv * a.cfw_rows : nullptr;
                }
                const Moments mo = feature_vector_moments8x2<kMode == 3 ? 2 : 4>(a.simmeasure, go, jj, D, sa, cw, a.cfw_rows, f0, f1, f2, s_w[3 * q], s_w[3 * q + 1], s_w[3 * q + 2]);
                if (go && jj == 0) s_w[3 * q] = mo.pr, s_w[3 * q + 1] = mo.va, s_w[3 * q + 2] = mo.vb, s_w[3 * kThreads + q] = mo.sum;  // (the weights have been used)
            }
            __syncthreads();
            // the divisions and square roots that end the measure: a lane per sample -- once per wavefront and round instead of once per pass
            // (they were 13 of the kernel's 76 us at ico4 / 32 features)
            if (tt >= 0) s_vals[kk * m.cap + j] = similarity_from_moments(a.simmeasure, D, Moments{s_w[3 * tid], s_w[3 * tid + 1], s_w[3 * tid + 2], s_w[3 * kThreads + tid]});
        }
    }
    MSM_STAMP(3);
    if (total == 0) __syncthreads();  // the barrier of the first round, for a run of empty bins
    __syncthreads();
    MSM_STAMP(4);

    // ---- the samples the direction table left open (0.2 %): the octree leaf's candidates, eight lanes per sample
    // (search_device.hpp: group8_find); what even that cannot decide (no candidate in the leaf: sibling leaves, nearest vertex)
    // is left to the tail kernel, which the host launches only when told to
#ifdef MSM_MOVE_SKIP_OPEN  // diagnostics (WRONG results): what the leaf search of the open samples costs the launch -- 1: no workgroup runs it, 2: one in 32 does
    const int npend = ((MSM_MOVE_SKIP_OPEN == 2 && (blockIdx.x & 31) == 0) || m.cap < 0) ? s_npend : 0;  // (cap < 0 never holds: keeps the code in)
#else
    const int npend = s_npend;
#endif
    // the strain of the 64 evaluations (a chain of some hundred dependent FP64 operations that only needs the proposed triangles) is
    // computed by the last wavefront here, while the first ones search for the open samples and then start on the similarities:
    // it used to follow the similarity in every evaluation's lane, 2 us of the 4.3 us last phase
    if (tid >= kThreads - 64) {
        const int h = tid - (kThreads - 64);
        if (h < 8 * ntrip && !(m.single && (h & 7))) {
            const double *gp = s_geo + 9 * h;
            const V3 rr[3] = {mk(gp[0], gp[1], gp[2]), mk(gp[3], gp[4], gp[5]), mk(gp[6], gp[7], gp[8])};
            s_strain[h] = s_flag[h] ? 0.0 : move_strain(a, s_frame + 5 * (h >> 3), rr);
        }
    }
#ifdef MSM_MOVE_TRACE
    if (m.trace && tid == 0) m.trace[8 * (size_t)blockIdx.x + 7] = (unsigned long long)npend;  // how many samples this workgroup had to search for
#endif
    for (int q0 = 0; q0 < npend; q0 += kThreads / 8) {  // workgroup-uniform
        const int q = q0 + (tid >> 3);
        const bool valid = q < npend;
        int kk = 0, j = 0, el = 0;
        V3 p = mk(0.0, 0.0, 0.0);
        if (valid) {
            const int s = s_pend[q];
            kk = fast_div(s, nslots, inv), j = s - kk * nslots;
            const size_t slot = (size_t)(s0 + j);
            el = (m.slot_tri[slot] - t0) * 8 + kk;
            p = moved_point(s_geo + 9 * el, m.slot_w[3 * slot], m.slot_w[3 * slot + 1], m.slot_w[3 * slot + 2]);
        }
        if (!__any(valid)) continue;
        const int found = group8_find(a.tree, valid, p, lane);  // the same in the eight lanes of a group
        double val = 0.0;
        if (kMode >= 2) {
            // the similarity over the features like the sampling rounds above: eight lanes per sample (one lane through 32 dimensions was
            // 14 of the kernel's 88 us at ico4: a third of the workgroups have an open sample and every one of them waited for that lane)
            const bool go = valid && found >= 0;
            const int D = a.D;
            const double *f0 = a.tfeat, *f1 = a.tfeat, *f2 = a.tfeat, *sa = a.sfeat_vm, *cw = nullptr;
            double wa = 0.0, wb = 0.0, wc = 0.0;
            if (go) {
                const TriRec &r = a.tree.rec[found];
                area_weights(rec_v0(r), rec_v1(r), rec_v2(r), p, wa, wb, wc);  // every lane: same instructions, same values
                const int sv = a.bin_idx[s0 + j];
                f0 = a.tfeat + (size_t)r.id[0] * D, f1 = a.tfeat + (size_t)r.id[1] * D, f2 = a.tfeat + (size_t)r.id[2] * D;
                sa = a.sfeat_vm + (size_t)sv * D;
                cw = a.cfw_vm ? a.cfw_vm + (size_t)sv * a.cfw_rows : nullptr;
            }
            val = feature_vector_similarity8x2(a.simmeasure, go, lane & 7, D, sa, cw, a.cfw_rows, f0, f1, f2, wa, wb, wc);
        } else if (valid && (lane & 7) == 0 && found >= 0) {
            val = ho_value_on(a, a.bin_idx[s0 + j], p, found);
        }
        if (valid && (lane & 7) == 0) {
            if (found >= 0) {
                s_vals[kk * m.cap + j] = val;
            } else {
                s_vals[kk * m.cap + j] = pending_value();
                s_flag[64 + el] = 1;
            }
        }
    }
    __syncthreads();  // the open samples' values and the strains are in LDS
    MSM_STAMP(5);

    // ---- one lane per evaluation: similarity in the reference's serial order + strain
    if (!ev) return;
    const int t = t0 + (tid >> 3), k = tid & 7, e = 8 * t + k;
    double *dst = m.out + (m.single ? t : e);
    if (s_flag[tid]) {
        *dst = MSM_FOLDING * a.lambda;
        return;
    }
    const int beg = s_bin[tid >> 3], n = s_bin[(tid >> 3) + 1] - beg;
    const double *vals = s_vals + k * m.cap + (beg - s0);
    if (s_flag[64 + tid]) {  // some point is still open: hand the evaluation and what is known of it to the tail kernel
        const unsigned at = atomicAdd(&m.defer_cnt[m.parity], 1u);
        m.defer_list[at] = (unsigned)e;
        double *gv = m.vals + (size_t)8 * beg + (size_t)k * n;
        for (int i = 0; i < n; ++i) gv[i] = vals[i];
        __hip_atomic_store(m.host_flags + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    *dst = move_likelihood(a, n, s_stat + 3 * (tid >> 3), s_wda + (beg - s0), m.slot_cw ? s_cw + (beg - s0) : nullptr, s_sf + (beg - s0), wmean, vals) + s_strain[tid];
    MSM_STAMP(6);
}

// The evaluations the main kernel could not finish: a wavefront each, eight lanes per open point (complete search), then the
// same reduction.  Launched by the host only when the main kernel said so (host_flags[1]).
template <bool kPacked>
__global__ __launch_bounds__(256) void k_ho_move_tail(CliqueArgs a, MoveArgs m, MoveLabels lab) {
    extern __shared__ __align__(16) double lds[];  // 4 wavefronts x bin_cap values
    const unsigned n = m.defer_cnt[m.parity];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane >> 3, sub = lane & 7;
    double *vals = lds + (size_t)wave * a.bin_cap;
    const unsigned stride = gridDim.x * 4;
    for (unsigned base = blockIdx.x * 4; base < n; base += stride) {  // workgroup-uniform
        const unsigned idx = base + wave;
        const bool have = idx < n;
        int t = 0, k = 0, id[3] = {0, 0, 0}, beg = 0, cnt = 0;
        V3 r[3];
        r[0] = r[1] = r[2] = mk(0.0, 0.0, 0.0);
        if (have) {
            const int e = (int)m.defer_list[idx];
            t = e >> 3, k = e & 7;
            proposed_triangle<kPacked>(a, m, lab, t, k, id, r);  // not folded: a folded proposal is never listed
            beg = a.bin_ptr[t], cnt = a.bin_ptr[t + 1] - beg;
        }
        const double g[9] = {r[0].x, r[0].y, r[0].z, r[1].x, r[1].y, r[1].z, r[2].x, r[2].y, r[2].z};
        double *gv = m.vals + (size_t)8 * beg + (size_t)k * cnt;
        for (int i0 = 0; i0 < cnt; i0 += 8) {  // wavefront-uniform: all lanes of a wavefront share the evaluation
            const int i = i0 + grp;
            const bool in = i < cnt;
            double v = in ? gv[i] : 0.0;
            const bool pend = in && is_pending(v);
            V3 p = mk(0.0, 0.0, 0.0);
            if (pend) {
                const size_t slot = (size_t)(beg + i);
                p = moved_point(g, m.slot_w[3 * slot], m.slot_w[3 * slot + 1], m.slot_w[3 * slot + 2]);
            }
            if (pend && sub == 0) {
                const int tt = find_closest_triangle(a.tree, p);
                if (tt < 0) {
                    raise_both(a, m, tt);
                    v = __longlong_as_double(0x7ff8000000000000ll);
                } else {
                    v = ho_value_on(a, a.bin_idx[beg + i], p, tt);
                }
            }
            if (in && sub == 0) vals[i] = v;
        }
        __syncthreads();
        if (have && lane == 0)
            m.out[m.single ? t : 8 * t + k] = move_likelihood(a, cnt, m.tri_stat + 3 * (size_t)t, m.slot_wda + beg, m.slot_cw ? m.slot_cw + beg : nullptr, m.slot_sf + beg,
                                               (a.absw[id[0]] + a.absw[id[1]] + a.absw[id[2]]) / 3.0, vals) +
                               move_strain(a, m.tri_frame + 5 * (size_t)t, r);
        __syncthreads();
    }
}

int launch_move_prepare(msm_ctx *ctx, const CliqueArgs &a, int nslots, int *slot_tri, double *slot_w, double *slot_sf, double *slot_cw, double *slot_wda,
                        double *tri_frame, double *tri_stat) {
    if (a.T <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_move_prepare, dim3((a.T + 255) / 256), dim3(256), 0, ctx->stream, a, slot_tri, slot_w, slot_sf, slot_cw, slot_wda, tri_frame, tri_stat);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

static_assert(sizeof(CliqueArgs) + sizeof(MoveArgs) + sizeof(MoveLabels) <= 4096, "the fusion move's arguments must fit the 4 KB kernel-argument segment");

static int move_mode(const CliqueArgs &a) {
    if (a.kind != MSM_COST_HO_MULTIVARIATE) return 0;
    if (!(a.sfeat_vm && a.D >= 12 && a.D <= 64 && a.D % 2 == 0 && (a.simmeasure == 1 || a.simmeasure == 2))) return 1;
    return a.D <= 32 ? 3 : 2;  // 3: two dimension pairs per lane instead of four
}
static MoveLabels g_no_labels;  // handed over (and never read) when the labeling comes as a device array

int launch_move(msm_ctx *ctx, const CliqueArgs &a, const MoveArgs &m, const MoveLabels *labels, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (a.T <= 0 || m.nblk <= 0) return MSM_OK;
    const int mode = move_mode(a);
    // 256 threads and two rounds of samples per workgroup.  Measured alternatives (D = 1 / D = 32 kernel time against 32 / 95 us):
    // 512 threads, all samples of the 8 triangles in flight at once (MSMHIP_MOVE_THREADS=512) 37 / 102; 4 triangles and one round per
    // workgroup 34.5 / 91, 2 triangles 46 / 107; the records of a wavefront's 64 first candidates fetched as a team through LDS
    // (neighbouring lanes reading neighbouring 16-byte pieces) 33 / 142.
    static const int threads = [] { const char *e = std::getenv("MSMHIP_MOVE_THREADS"); return e && std::atoi(e) == 512 ? 512 : 256; }();
    const dim3 grid((unsigned)(8 * ((m.nblk + 7) / 8))), block(threads);
    const size_t lds = sizeof(double) * (64 * 9 + 8 * (size_t)m.cap + 3 * (size_t)m.cap + 24 + 40 + 64 + (mode >= 2 ? 4 * threads : 0)) +
                       sizeof(int) * (128 + 8 * (size_t)m.cap + (mode >= 2 ? 4 * threads : 0) + 16);
    if (lds > 64 * 1024) return fail(MSM_ERR_CAPACITY, "fusion move: %d bin slots per workgroup do not fit LDS", m.cap);
    const MoveLabels &lab = labels ? *labels : g_no_labels;
    if (ev_start) MSM_HIP(hipEventRecord(ev_start, ctx->stream));
#define MSM_MOVE_LAUNCH(PACKED)                                                                                     \
    do {                                                                                                            \
        if (threads == 512) {                                                                                       \
            if (mode == 0) hipLaunchKernelGGL((k_ho_move<PACKED, 0, 512>), grid, block, lds, ctx->stream, a, m, lab);     \
            else if (mode == 1) hipLaunchKernelGGL((k_ho_move<PACKED, 1, 512>), grid, block, lds, ctx->stream, a, m, lab); \
            else if (mode == 2) hipLaunchKernelGGL((k_ho_move<PACKED, 2, 512>), grid, block, lds, ctx->stream, a, m, lab); \
            else hipLaunchKernelGGL((k_ho_move<PACKED, 3, 512>), grid, block, lds, ctx->stream, a, m, lab);                \
        } else {                                                                                                    \
            if (mode == 0) hipLaunchKernelGGL((k_ho_move<PACKED, 0, 256>), grid, block, lds, ctx->stream, a, m, lab);     \
            else if (mode == 1) hipLaunchKernelGGL((k_ho_move<PACKED, 1, 256>), grid, block, lds, ctx->stream, a, m, lab); \
            else if (mode == 2) hipLaunchKernelGGL((k_ho_move<PACKED, 2, 256>), grid, block, lds, ctx->stream, a, m, lab); \
            else hipLaunchKernelGGL((k_ho_move<PACKED, 3, 256>), grid, block, lds, ctx->stream, a, m, lab);                \
        }                                                                                                           \
    } while (0)
    if (labels) MSM_MOVE_LAUNCH(true);
    else MSM_MOVE_LAUNCH(false);
#undef MSM_MOVE_LAUNCH
    MSM_HIP(hipGetLastError());
    if (ev_stop) MSM_HIP(hipEventRecord(ev_stop, ctx->stream));
    return MSM_OK;
}

int launch_move_tail(msm_ctx *ctx, const CliqueArgs &a, const MoveArgs &m, const MoveLabels *labels) {
    const size_t lds_tail = sizeof(double) * 4 * (size_t)std::max(a.bin_cap, 1);
    if (lds_tail > 64 * 1024) return fail(MSM_ERR_CAPACITY, "fusion move: bins of %d points do not fit LDS", a.bin_cap);
    const MoveLabels &lab = labels ? *labels : g_no_labels;
    if (labels) hipLaunchKernelGGL((k_ho_move_tail<true>), dim3(64), dim3(256), lds_tail, ctx->stream, a, m, lab);
    else hipLaunchKernelGGL((k_ho_move_tail<false>), dim3(64), dim3(256), lds_tail, ctx->stream, a, m, lab);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

}  // namespace msm
