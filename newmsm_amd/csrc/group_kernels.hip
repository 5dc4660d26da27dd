// group_kernels.hip -- groupwise (gMSM) kernels: the per-label rigid rotation of a subject's data mesh
// (DiscreteGroupModel::get_patch_data, M/DiscreteGroupModel.cpp:99-103), the inter-subject patch similarity
// (DiscreteGroupCostFunction::computePairwiseCost, M/DiscreteGroupCostFunction.cpp:54-98) and the per-subject strain
// triplet (:26-52).
#include "kernels.hpp"
#include "search_device.hpp"
#include "similarity_device.hpp"
#include "strain_device.hpp"

namespace msm {

namespace {

__device__ __forceinline__ void raise_status(int *status, int code) { atomicMin(status, code); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum over the kLanes lanes (64, or an aligned half of the wavefront) that share a query
template <int kLanes>
__device__ __forceinline__ double lanes_sum(double v) {
#pragma unroll
    for (int off = kLanes / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sums of N values over the 32 lanes of a half wavefront at once.  The plain butterfly costs five exchanges per value; here the
// first three steps also halve the number of values a lane carries (at the xor-16 step the lower lanes go on with the even values
// and hand the odd ones over, and so on), so N values cost ceil(N/2) + ceil(N/4) + ceil(N/8) + 2 exchanges plus N broadcasts -- 13
// instead of 25 for the five sums of a two-row correlation.  Every sum pairs the same lanes in the same order as the butterfly
// (16, 8, 4, 2, 1), so the results are the butterfly's bit for bit.  All 64 lanes call it; `lane` is the lane within the half.
// half_reduce leaves value k, complete, in the lanes whose bits 16 / 8 / 4 are bits 0 / 1 / 2 of k (four lanes each); half_get
// fetches one.  Keeping the sums where they land lets the per-query scalar arithmetic that follows (divisions by the sum of
// weights, square roots) run ONCE with a different value in every lane group, instead of once per value in all lanes.
// v of lane (lane ^ kOff): a DPP move within a row of sixteen lanes (similarity_device.hpp), a shuffle through the LDS pipe across rows
template <int kOff>
__device__ __forceinline__ double xor_exchange(double v) {
    if constexpr (kOff >= 16) return __shfl_xor(v, kOff, 64);
    else return dpp_xor<kOff>(v);
}
template <int N, int kLanes = 32>
__device__ __forceinline__ double half_reduce(const double (&v)[N], int lane) {
    static_assert(N >= 1 && N <= 8, "at most eight values");
    static_assert(kLanes == 32 || kLanes == 16, "half or quarter of a wavefront");
    constexpr int N1 = (N + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2, T = kLanes / 2;
    static_assert(N3 == 1, "three halving steps");
    double w[N1], x[N2], y;
    const bool u1 = lane & T, u2 = lane & (T / 2), u3 = lane & (T / 4);
#pragma unroll
    for (int j = 0; j < N1; ++j) {
        const double e = v[2 * j], o = 2 * j + 1 < N ? v[2 * j + 1] : 0.0;
        w[j] = (u1 ? o : e) + xor_exchange<T>(u1 ? e : o);
    }
#pragma unroll
    for (int j = 0; j < N2; ++j) {
        const double e = w[2 * j], o = 2 * j + 1 < N1 ? w[2 * j + 1] : 0.0;
        x[j] = (u2 ? o : e) + xor_exchange<T / 2>(u2 ? e : o);
    }
    {
        const double e = x[0], o = N2 > 1 ? x[N2 - 1] : 0.0;
        y = (u3 ? o : e) + xor_exchange<T / 4>(u3 ? e : o);
    }
    if constexpr (T / 8 >= 2) y += xor_exchange<2>(y);
    if constexpr (T / 8 >= 1) y += xor_exchange<1>(y);
    return y;
}
// which value a lane holds after half_reduce, and a lane that holds value k (kLanes == 16: the same with the lane bits 8 / 4 / 2)
template <int kLanes = 32>
__device__ __forceinline__ int half_slot(int lane) {
    constexpr int T = kLanes / 2;
    return ((lane / T) & 1) | (((lane / (T / 2)) & 1) << 1) | (((lane / (T / 4)) & 1) << 2);
}
template <int kLanes = 32>
__device__ __forceinline__ int half_lane_of(int k) {
    constexpr int T = kLanes / 2;
    return ((k & 1) * T) | (((k >> 1) & 1) * (T / 2)) | (((k >> 2) & 1) * (T / 4));
}
template <int kLanes = 32>
__device__ __forceinline__ double half_get(double y, int k) { return __shfl(y, ((threadIdx.x & 63) & ~(kLanes - 1)) | half_lane_of<kLanes>(k), 64); }
template <int N>
__device__ __forceinline__ void half_sums(double (&v)[N], int lane) {
    const double y = half_reduce<N>(v, lane);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = half_get(y, k);
}

constexpr int kGroupStage = 256;  // ids of patch B staged in LDS per wavefront

// a pointer the compiler knows to point into device memory (address space 1)
template <class T>
using GlobalPtr = const T __attribute__((address_space(1))) *;
typedef double Dbl2 __attribute__((ext_vector_type(2)));  // (a builtin vector: loadable through an address-space pointer, which HIP's double2 class is not)
template <class T>
__device__ __forceinline__ GlobalPtr<T> as_global(const T *p) {
    return (GlobalPtr<T>)p;
}

}  // namespace

// every data vertex v is moved to estimate_rotation_matrix(centre, v) * label: "rigid rotation" of the mesh by the label
__global__ __launch_bounds__(256) void k_rotate_to_label(const double *__restrict__ xyz, int V, V3 centre, V3 label, double *__restrict__ out,
                                                          size_t stride, int *status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    double R[9];
    if (!rotation_matrix(centre, mk(xyz[i], xyz[V + i], xyz[2 * V + i]), R)) raise_status(status, MSM_ERR_ROTATION);
    const V3 p = rotate(R, label);
    out[i] = p.x;
    out[stride + i] = p.y;
    out[2 * stride + i] = p.z;
}

// The same for ALL labels of a set-up in one launch: the rotation of a vertex does not depend on the label (estimate_rotation_matrix(centre, vertex)), only the
// vector it is applied to does -- k_rotate_to_label per label computed it L - 1 times.  out: 3 x (L * V), x of label 0, x of label 1, ..., y of label 0, ...
// (stride = L * V); label 0 is the centre of the sampling grid: the coordinates as they are (M/DiscreteGroupModel.cpp:100: "if (label > 0)").
// rot9 != nullptr: the V matrices come from the host (msm_group_set_rotation_mode: the host's libm, as the reference computes them) and are only applied here.
__global__ __launch_bounds__(256) void k_rotate_to_labels(const double *__restrict__ xyz, int V, V3 centre, const double *__restrict__ labels, int L,
                                                           const double *__restrict__ rot9, double *__restrict__ out, size_t stride, int *status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const V3 p = mk(xyz[i], xyz[V + i], xyz[2 * (size_t)V + i]);
    double R[9];
    if (rot9) {
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = rot9[9 * (size_t)i + k];
    } else if (!rotation_matrix(centre, p, R)) {
        raise_status(status, MSM_ERR_ROTATION);
    }
    out[i] = p.x, out[stride + i] = p.y, out[2 * stride + i] = p.z;
    for (int l = 1; l < L; ++l) {
        const V3 q = rotate(R, mk(labels[l], labels[L + l], labels[2 * (size_t)L + l]));
        const size_t at = (size_t)l * V + i;
        out[at] = q.x, out[stride + at] = q.y, out[2 * stride + at] = q.z;
    }
}

int launch_rotate_to_labels(msm_ctx *ctx, const double *d_xyz, int V, const double centre[3], const double *d_labels3, int L, const double *d_rot9, double *d_out,
                            size_t stride) {
    if (V <= 0 || L <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_rotate_to_labels, dim3((V + 255) / 256), dim3(256), 0, ctx->stream, d_xyz, V, mk(centre[0], centre[1], centre[2]), d_labels3, L, d_rot9, d_out,
                       stride, ctx->d_status);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_rotate_to_label(msm_ctx *ctx, const double *d_xyz, int V, const double centre[3], const double label[3], double *d_out, size_t stride) {
    hipLaunchKernelGGL(k_rotate_to_label, dim3((V + 255) / 256), dim3(256), 0, ctx->stream, d_xyz, V, mk(centre[0], centre[1], centre[2]),
                       mk(label[0], label[1], label[2]), d_out, stride ? stride : (size_t)V, ctx->d_status);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

// One wavefront per query.  Patch A = template vertices in range of (subject A, control point, label A), ascending;
// the lanes take A's entries, look each up in B's list by binary search (std::map::find), and the similarity of
// the two subjects' resampled features over the intersection is reduced per feature dimension with shuffles.
// kDice (DICE / genDICE): the common entries of one feature dimension are first packed, in patch order, into the
// wavefront's LDS rows (2 x patch_cap doubles) and thresholded there by rank counting (similarity_device.hpp).
template <bool kDice, int kLanes>
__global__ __launch_bounds__(256) void k_group_pairwise(GroupArgs a, const int *__restrict__ qp, const int *__restrict__ qa,
                                                         const int *__restrict__ qb, int n, double *__restrict__ out) {
    static_assert(kLanes == 64 || ((kLanes == 32 || kLanes == 16) && !kDice), "DICE uses whole wavefronts");
    constexpr int kPerBlock = 256 / kLanes;
    // workgroups are dealt to the 8 XCDs in turn: give each XCD (its own L2) a contiguous eighth of the launch's queries
    int blk = blockIdx.x;
    {
        const int per = gridDim.x >> 3;
        if (blk < (per << 3)) blk = (blk & 7) * per + (blk >> 3);
    }
    const int slot = threadIdx.x / kLanes, q = blk * kPerBlock + slot, lane = threadIdx.x & (kLanes - 1);
    if (q >= n) return;
    extern __shared__ double s_common[];
    int pair, la, lb, ga_known = -1, gb_known = -1;
    size_t at = (size_t)q;
    double *keep = nullptr;  // where this query's cost is kept for the next label step
    if (a.move_labeling) {  // Fusion's pair_data[pair].buffer[k], I/Fusion/Fusion.h:170-173: k = 2 * (A takes the label) + (B takes it)
        const int e = q + a.move_offset;
        int idx = a.move_combos == 0 ? e >> 2 : (a.move_combos == 1 ? e : (a.move_combos == 2 ? e / 3 : e >> 1));
        int k = a.move_combos == 0 ? (e & 3) : (a.move_combos == 1 ? 0 : (a.move_combos == 2 ? 1 + (e - 3 * idx) : 1 + (e & 1)));
        if (a.move_order4 && a.move_count > 0 && a.move_combos != 1) {  // four positions at a time, combination by combination (GroupArgs::move_first)
            const int m = a.move_combos == 0 ? 4 : (a.move_combos == 2 ? 3 : 2);
            const int r = e - m * a.move_first, full = a.move_count & ~3;
            int ks;
            if (r < m * full) {
                const int c = r / (4 * m), j = r - c * 4 * m;
                idx = a.move_first + 4 * c + (j & 3), ks = j >> 2;
            } else {
                const int r2 = r - m * full, p = r2 / m;
                idx = a.move_first + full + p, ks = r2 - p * m;
            }
            k = a.move_combos == 0 ? ks : 1 + ks;
        }
        int nodeA, nodeB;
        if (a.move_order4) {  // (round 5: one 16-byte load for the position's pair and its nodes)
            const int4 o = a.move_order4[idx];
            pair = o.x, nodeA = o.y, nodeB = o.z;
        } else {
            pair = a.move_order ? a.move_order[idx] : idx;
            nodeA = a.pairs[2 * pair], nodeB = a.pairs[2 * pair + 1];
        }
        if (a.move_order) at = 4 * (size_t)(pair - a.move_base) + k;
        ga_known = nodeA, gb_known = nodeB;
        const int curA = a.move_labeling[nodeA], curB = a.move_labeling[nodeB];
        if (k == 0 && a.move_e00) {
            if (a.move_prev && a.move_prev[nodeA] == curA && a.move_prev[nodeB] == curB) {  // same patches as in the previous step
                if (lane == 0) out[at] = a.move_e00[pair];
                return;
            }
            keep = a.move_e00 + pair;
        }
        if (k == 3 && a.move_e11) keep = a.move_e11 + (pair - a.move_base);
        // A combination whose proposed label IS the node's current one repeats (current, current): taken from the pass that has just evaluated or kept it
        // (same kernel, same inputs: the same bits).  A nineteenth of the evaluations with uniformly random labels; most of them in a registration that is
        // converging, where the proposed label is the current one of most nodes.
        if (k != 0 && a.move_e00 && (a.move_combos == 2 || a.move_combos == 3) && (!(k & 2) || curA == a.move_label) && (!(k & 1) || curB == a.move_label)) {
            if (lane == 0) {
                const double c = a.move_e00[pair];
                out[at] = c;
                if (keep) *keep = c;
            }
            return;
        }
        la = (k & 2) ? a.move_label : curA;
        lb = (k & 1) ? a.move_label : curB;
    } else {
        pair = qp[q], la = qa[q], lb = qb[q];
    }
    const int ga = ga_known >= 0 ? ga_known : a.pairs[2 * pair], gb = gb_known >= 0 ? gb_known : a.pairs[2 * pair + 1];
    const int sa = ga / a.N, sb = gb / a.N, na = ga - sa * a.N, nb = gb - sb * a.N;
    // The per-subject arrays are reached through pointer tables in device memory.  A pointer LOADED from memory is a generic one to the compiler, and every
    // access through it a FLAT instruction (the vector-memory path plus the LDS path's wait counter: the 20 feature gathers and 10 list loads of a query);
    // they are device memory, and say so (round 5: global_load instead of flat_load).
    // Round 5: with the patch directory (GroupArgs::dir, built with the value copies) one 32-byte record per side says where the patch's ids and values lie and
    // how long it is -- the pointer tables and the row offsets were two more dependent round trips in front of every query's data.
    int ba, bb, cntA, cntB;
    GlobalPtr<int> ia, ib;
    GlobalPtr<double> dirA = nullptr, dirB = nullptr;
    if (a.dir) {
        const GroupPatchRef ra = a.dir[(size_t)ga * a.L + la], rb = a.dir[(size_t)gb * a.L + lb];
        ia = as_global(ra.ids), ib = as_global(rb.ids);
        dirA = as_global(ra.vals), dirB = as_global(rb.vals);
        ba = ra.begin, bb = rb.begin, cntA = ra.count, cntB = rb.count;
    } else {
        const GlobalPtr<int> pa = as_global(a.pptr[sa]), pb = as_global(a.pptr[sb]);
        ba = pa[na * a.L + la], bb = pb[nb * a.L + lb];
        cntA = pa[na * a.L + la + 1] - ba, cntB = pb[nb * a.L + lb + 1] - bb;
        ia = as_global(a.pidx[sa]) + ba, ib = as_global(a.pidx[sb]) + bb;
    }
    const GlobalPtr<double> FA = as_global(a.F[(size_t)sa * a.L + la]), FB = as_global(a.F[(size_t)sb * a.L + lb]);
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    // B's ids go to LDS first when they fit (patches hold ~65 entries at ico6 / ico4): the binary search below is a chain
    // of dependent loads, and seven round trips to memory per query were most of this kernel's time
    // (+ 8 words per slot: the slots of a wavefront's queries start 8 banks apart -- their lanes walk lists of similar length in lockstep and would otherwise
    // read the same index of four lists, i.e. four addresses in one bank, at every step of the search)
    __shared__ int s_ids[kPerBlock][kGroupStage + 8];
    int *stage = s_ids[slot];
    // Staged up to the next of 64 / 128 / 256 entries, the tail filled with INT_MAX: the fast path's search then walks a power of two with no bounds to check.
    const bool staged = cntB <= kGroupStage;
    const int padded = cntB <= 64 ? 64 : (cntB <= 128 ? 128 : 256);
    // The usual case -- patches of up to 5 (4) entries per lane, one or two feature rows, correlation or SSD -- keeps everything in registers (below).  Its
    // loads are requested HERE, all at once and unconditionally (indices clamped into the patch instead of branched around): patch A's ids, patch A's
    // values from the entry-major copy beside them (GroupArgs::pval) and the first 64 ids of patch B.  Round 5: B's ids used to be fetched and written to
    // LDS one round after the other (load, wait, write, four times), and A's ids only after that -- five dependent memory phases in front of the search
    // where one does; A's values were ten gathers by vertex id behind the search.
    constexpr int kRounds = kLanes == 16 ? kPairSmallPatch / 16 : 4;  // entries per lane kept in registers: patches of up to 80 / 128 entries (6 and 8
                                                                      // rounds of sixteen lanes: 89 / 109 registers, 10.2 / 12.1 ms per label step against 9.5)
    constexpr int kQ = kLanes == 64 ? 32 : kLanes;                    // (the fast path is compiled for 32 and 16 lanes per query)
    const bool fast = !kDice && staged && cntA <= kRounds * kLanes && a.D <= 2 && a.simmeasure != 4 && a.simmeasure != 5;
    const bool compact = fast && a.pval != nullptr;
    const GlobalPtr<double> PA = !compact ? GlobalPtr<double>(nullptr) : (a.dir ? dirA : as_global(a.pval[sa]) + (size_t)a.D * ba);
    const GlobalPtr<double> PB = !compact ? GlobalPtr<double>(nullptr) : (a.dir ? dirB : as_global(a.pval[sb]) + (size_t)a.D * bb);
    int id[kRounds];
    double pa_[kRounds][2];
    if (fast) {
        const int lastA = max(cntA, 1) - 1;
#pragma unroll
        for (int r = 0; r < kRounds; ++r) {
            const int i = lane + r * kLanes, ic = min(i, lastA);
            const int v = ia[ic];
            id[r] = i < cntA ? v : -1;
            pa_[r][0] = pa_[r][1] = 0.0;
            if (compact) {
                if (a.D == 2) {
                    const Dbl2 pv = *reinterpret_cast<const __attribute__((address_space(1))) Dbl2 *>(PA + 2 * (size_t)ic);
                    pa_[r][0] = pv.x, pa_[r][1] = pv.y;
                } else {
                    pa_[r][0] = PA[ic];
                }
            }
        }
    }
    if (staged) {
        const int lastB = max(cntB, 1) - 1;
        int bid[64 / kLanes];
#pragma unroll
        for (int r = 0; r < 64 / kLanes; ++r) bid[r] = ib[min(lane + r * kLanes, lastB)];  // (the first 64 without a loop: most patches end there or in the next 64)
#pragma unroll
        for (int r = 0; r < 64 / kLanes; ++r) {
            const int i = lane + r * kLanes;
            stage[i] = i < cntB ? bid[r] : 0x7fffffff;
        }
        for (int i = lane + 64; i < padded; i += kLanes) stage[i] = i < cntB ? ib[i] : 0x7fffffff;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int *fb = staged ? (const int *)stage : (const int *)ib;  // (generic: the general path below reads it either way)
    if constexpr (!kDice) {
        // Everything of a query in registers: ids loaded once, the values of both rows for both passes at hand.  The general code below reads ids and values
        // again in every pass of every row: thirteen dependent round trips to memory per query, and this kernel is bound by exactly that latency (64 % of its
        // wave cycles were waits; its L2 misses did not matter).  Same per-lane accumulation order, same shuffles: the results are bit-identical to the
        // general code's.  B's ids are read through `stage`, a pointer the compiler knows to be LDS (through `fb`, which may point either way, every step
        // of the search was a FLAT load).
        if (fast) {
            bool mem[kRounds];
            // The last entry of B that is <= the round's id, all rounds in lockstep (independent LDS reads per step instead of one dependent chain per
            // round): steps of padded / 2 ... 1 over the padded list, no bounds and no per-round branches (an unused round carries id -1 and stays at
            // entry 0) -- round 5: the lower_bound it replaces cost two more look-ups per round and a scalar branch per round and step, a quarter of the
            // kernel's vector and nearly all of its scalar instructions.  Membership is an equality either way: the same common set.
            int base[kRounds];
#pragma unroll
            for (int r = 0; r < kRounds; ++r) base[r] = 0;
            for (int step = padded >> 1; step > 0; step >>= 1) {  // the same trip count for the lanes of a query
#pragma unroll
                for (int r = 0; r < kRounds; ++r) {
                    const int t = base[r] + step;
                    base[r] = stage[t] <= id[r] ? t : base[r];
                }
            }
#pragma unroll
            for (int r = 0; r < kRounds; ++r) mem[r] = id[r] >= 0 && stage[base[r]] == id[r];
            // size of the intersection: the set bits of the rounds' ballots within this half wavefront
            int ncommon = 0;
            {
                const unsigned long long halfmask = (kQ == 32 ? 0xffffffffull : 0xffffull) << ((threadIdx.x & 63) & ~(kQ - 1));
#pragma unroll
                for (int r = 0; r < kRounds; ++r) ncommon += __popcll(__ballot(mem[r]) & halfmask);
            }
            double cost = 0.0;
            if (ncommon == 0) {
                cost = nan;
            } else {
                double va[kRounds][2], vb[kRounds][2], w[kRounds];
#pragma unroll
                for (int r = 0; r < kRounds; ++r) {
                    w[r] = mem[r] ? 1.0 : 0.0;  // an entry outside the intersection takes part with weight 0 and values 0: sums unchanged, no branches
                    va[r][0] = va[r][1] = vb[r][0] = vb[r][1] = 0.0;
                    if (mem[r]) {
                        if (compact) {  // A's values are in registers; B's at the position the search ended on (one 16-byte load for two rows)
                            va[r][0] = pa_[r][0], va[r][1] = pa_[r][1];
                            if (a.D == 2) {
                                const Dbl2 v = *reinterpret_cast<const __attribute__((address_space(1))) Dbl2 *>(PB + 2 * (size_t)base[r]);
                                vb[r][0] = v.x, vb[r][1] = v.y;
                            } else {
                                vb[r][0] = PB[base[r]];
                            }
                        } else {
#pragma unroll
                            for (int d = 0; d < 2; ++d)
                                if (d < a.D) {
                                    va[r][d] = FA[(size_t)d * a.Vt + id[r]];
                                    vb[r][d] = FB[(size_t)d * a.Vt + id[r]];
                                }
                        }
                    }
                    if (mem[r] && a.mask) w[r] = fabs(a.mask[id[r]]);
                }
                const bool two = a.D == 2;
                const int row = (lane / (kQ / 2)) & 1;  // the tail of feature row 0 runs in the lower half of the query's lanes, of row 1 in the upper
                double c_row;                     // this lane's row's similarity
                if (a.simmeasure == 2) {  // sparsesimkernel::corr, M/similarities.cpp:129-158, both feature rows side by side
                    double s1[5] = {0.0, 0.0, 0.0, 0.0, 0.0};  // sum of weights, then weighted sums of A and B per row
                    if (a.mask) {
#pragma unroll
                        for (int r = 0; r < kRounds; ++r) {
                            s1[0] += w[r];
#pragma unroll
                            for (int d = 0; d < 2; ++d) {
                                s1[1 + 2 * d] += w[r] * va[r][d];
                                s1[2 + 2 * d] += w[r] * vb[r][d];
                            }
                        }
                    } else {  // weights 1 (in the intersection) or 0 with values 0 (outside): w * v IS v, bit for bit -- twenty multiplications less per lane
#pragma unroll
                        for (int r = 0; r < kRounds; ++r) {
                            s1[0] += w[r];
#pragma unroll
                            for (int d = 0; d < 2; ++d) {
                                s1[1 + 2 * d] += va[r][d];
                                s1[2 + 2 * d] += vb[r][d];
                            }
                        }
                    }
                    // every lane group divides ITS sum by the sum of weights: one division for the four means
                    double y = half_reduce<5, kQ>(s1, lane);
                    const double sw = half_get<kQ>(y, 0);
                    if (sw > 0.0) y /= sw;
                    const double ma[2] = {half_get<kQ>(y, 1), half_get<kQ>(y, 3)}, mb[2] = {half_get<kQ>(y, 2), half_get<kQ>(y, 4)};
                    double s2[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // per row: products, variance of A, variance of B
#pragma unroll
                    for (int r = 0; r < kRounds; ++r)
                        {
#pragma unroll
                            for (int d = 0; d < 2; ++d) {
                                const double da = va[r][d] - ma[d], db = vb[r][d] - mb[d];
                                s2[3 * d] += w[r] * da * db;
                                s2[3 * d + 1] += w[r] * da * da;
                                s2[3 * d + 2] += w[r] * db * db;
                            }
                        }
                    // likewise one division for the six second moments and one square root for the four variances
                    double z = half_reduce<6, kQ>(s2, lane);
                    if (sw > 0.0) z /= sw;
                    const double rt = sqrt(z);  // of a product sum in the lanes that hold one: not used
                    const int src = (threadIdx.x & 63) & ~(kQ - 1);
                    const double pr = __shfl(z, src | half_lane_of<kQ>(3 * row), 64);
                    const double sa = __shfl(rt, src | half_lane_of<kQ>(3 * row + 1), 64), sb = __shfl(rt, src | half_lane_of<kQ>(3 * row + 2), 64);
                    // sqrt(x) == 0 exactly when x == 0: the reference's test on the variances
                    const double rr = (sa == 0.0 || sb == 0.0) ? 0.0 : pr / (sa * sb);
                    c_row = 1 - (1 + rr) * 0.5;
                } else {  // sparsesimkernel::SSD, :179-188
                    double s1[2] = {0.0, 0.0};
#pragma unroll
                    for (int r = 0; r < kRounds; ++r)
                        {
#pragma unroll
                            for (int d = 0; d < 2; ++d) {
                                const double df = va[r][d] - vb[r][d];
                                s1[d] += w[r] * df * df;
                            }
                        }
                    const double y = half_reduce<2, kQ>(s1, lane);  // row 0 in the lower half of the query's lanes, row 1 in the upper
                    c_row = sqrt(y) / ncommon;
                }
                // cost = (row 0 [+ row 1]) / D, the rows added in the reference's order
                const double c0 = __shfl(c_row, ((threadIdx.x & 63) & ~(kQ - 1)), 64), c1 = __shfl(c_row, ((threadIdx.x & 63) & ~(kQ - 1)) | (kQ / 2), 64);
                cost = two ? c0 + c1 : c0;
                if (two) cost *= 0.5;  // x / 2 exactly (the division is a dozen instructions)
            }
            if (a.fixnan && cost != cost) cost = 1e7;  // FIX_NAN, M/reg_tools.h:31
            if (lane == 0) {
                out[at] = cost;
                if (keep) *keep = cost;
            }
            return;
        }
    }
    // membership of A's entries in B (std::map::find, M/DiscreteGroupCostFunction.cpp:75-78), kept as a bit per (lane, round) for the first 64
    // rounds (64 x kLanes entries: 1024 with a quarter wavefront per query); the entries of a larger patch are looked up again in every pass --
    // the reference has no limit on a patch's size and neither has this path (DICE packs the common entries into LDS: 64 lanes, 4096 entries)
    const auto in_b = [&](int id) {
        int lo = 0, hi = cntB;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (fb[mid] < id) lo = mid + 1;
            else hi = mid;
        }
        return lo < cntB && fb[lo] == id;
    };
    unsigned long long member = 0ull;
    int common = 0;
    for (int r = 0, i = lane; i < cntA; i += kLanes, ++r)
        if (in_b(ia[i])) {
            if (r < 64) member |= 1ull << r;
            ++common;
        }
    const auto is_member = [&](int r, int i) { return r < 64 ? (member >> r & 1ull) != 0 : in_b(ia[i]); };
    if (kDice && cntA > 64 * 64) raise_status(a.status, MSM_ERR_CAPACITY);
    const int ncommon = (int)lanes_sum<kLanes>((double)common);
    double cost = 0.0;
    if (ncommon == 0) {
        cost = nan;  // the reference indexes an empty vector here (undefined behaviour)
    } else {
        for (int d = 0; d < a.D; ++d) {
            const GlobalPtr<double> A = FA + (size_t)d * a.Vt, B = FB + (size_t)d * a.Vt;
            double c;
            if constexpr (kDice) {  // sparsesimkernel::DICE / genDICE, M/similarities.cpp:201-253 (weights unused)
                double *SA = s_common + (size_t)slot * 2 * a.patch_cap, *SB = SA + a.patch_cap;
                int base = 0;
                for (int r = 0, i0 = 0; i0 < cntA; i0 += 64, ++r) {
                    const int i = i0 + lane;
                    const bool m = i < cntA && (member >> r & 1ull);
                    const unsigned long long bal = __ballot(m);
                    if (m) {
                        const int pos = base + __popcll(bal & ((1ull << lane) - 1)), id = ia[i];
                        SA[pos] = A[id];
                        SB[pos] = B[id];
                    }
                    base += __popcll(bal);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                c = patch_dice(SA, SB, ncommon, lane, a.simmeasure, a.percentile);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else if (a.simmeasure == 2) {  // sparsesimkernel::corr, M/similarities.cpp:129-158
                double sw = 0, ma = 0, mb = 0;
                for (int r = 0, i = lane; i < cntA; i += kLanes, ++r)
                    if (is_member(r, i)) {
                        const int id = ia[i];
                        const double w = a.mask ? fabs(a.mask[id]) : 1.0;
                        sw += w;
                        ma += w * A[id];
                        mb += w * B[id];
                    }
                sw = lanes_sum<kLanes>(sw);
                ma = lanes_sum<kLanes>(ma);
                mb = lanes_sum<kLanes>(mb);
                if (sw > 0.0) {
                    ma /= sw;
                    mb /= sw;
                }
                double pr = 0, va = 0, vb = 0;
                for (int r = 0, i = lane; i < cntA; i += kLanes, ++r)
                    if (is_member(r, i)) {
                        const int id = ia[i];
                        const double w = a.mask ? fabs(a.mask[id]) : 1.0, da = A[id] - ma, db = B[id] - mb;
                        pr += w * da * db;
                        va += w * da * da;
                        vb += w * db * db;
                    }
                pr = lanes_sum<kLanes>(pr);
                va = lanes_sum<kLanes>(va);
                vb = lanes_sum<kLanes>(vb);
                if (sw > 0.0) {
                    pr /= sw;
                    va /= sw;
                    vb /= sw;
                }
                const double rr = (va == 0.0 || vb == 0.0) ? 0.0 : pr / (sqrt(va) * sqrt(vb));
                c = 1 - (1 + rr) * 0.5;
            } else {  // sparsesimkernel::SSD, :179-188
                double pr = 0;
                for (int r = 0, i = lane; i < cntA; i += kLanes, ++r)
                    if (is_member(r, i)) {
                        const int id = ia[i];
                        const double w = a.mask ? fabs(a.mask[id]) : 1.0, df = A[id] - B[id];
                        pr += w * df * df;
                    }
                pr = lanes_sum<kLanes>(pr);
                c = sqrt(pr) / ncommon;
            }
            cost += c;
        }
        cost /= a.D;
    }
    if (a.fixnan && cost != cost) cost = 1e7;  // FIX_NAN, M/reg_tools.h:31
    if (lane == 0) {
        out[at] = cost;
        if (keep) *keep = cost;
    }
}

__global__ __launch_bounds__(128) void k_group_triplet(GroupArgs a, const int *__restrict__ qt, const int *__restrict__ qa,
                                                        const int *__restrict__ qb, const int *__restrict__ qc, int n, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int t, lab[3];
    if (a.move_labeling) {  // triplet_data[t].buffer[k], k = 000..111, I/Fusion/Fusion.h:188-195
        const int e = i + a.move_offset;
        t = e >> 3;
        for (int k = 0; k < 3; ++k) lab[k] = (e >> (2 - k) & 1) ? a.move_label : a.move_labeling[a.triplets[3 * t + k]];
    } else {
        t = qt[i];
        lab[0] = qa[i], lab[1] = qb[i], lab[2] = qc[i];
    }
    const int s = t / a.Tc;
    V3 r[3], cur[3], org[3];
    for (int k = 0; k < 3; ++k) {
        const int gid = a.triplets[3 * t + k], v = gid - s * a.N;
        const double *m = a.moved + ((size_t)gid * a.L + lab[k]) * 3;
        r[k] = mk(m[0], m[1], m[2]);
        const double *c = a.cp + (size_t)s * 3 * a.N, *o = a.orig + (size_t)s * 3 * a.N;
        cur[k] = mk(c[v], c[a.N + v], c[2 * a.N + v]);
        org[k] = mk(o[v], o[a.N + v], o[2 * a.N + v]);
    }
    double cost;
    if (dot(tri_normal(r[0], r[1], r[2]), tri_normal(cur[0], cur[1], cur[2])) < 0.0) {
        cost = MSM_FOLDING;
    } else {
        const double e = triangular_strain(org, r, a.mu, a.kappa, a.k_exp);
        cost = (a.fixnan && e != e) ? 1e7 : a.subcorr * a.lambda * pow_exp(e, a.rexp);
    }
    out[i] = cost;
}

// ------------------------------------------------------------------------------------------------
// the parts of setupCostFunction that need every subject's control grid (identical on every rank)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ DevTree forest_tree_dev(const ForestDev &f, int b) {
    DevTree t;
    t.node = f.node + (size_t)b * f.s_node;
    t.parent = f.parent + (size_t)b * f.s_node;
    t.leaf_tri = f.leaf_tri + (size_t)b * f.s_leaf;
    t.cone = f.cone + (size_t)b * f.s_leaf;
    t.rec = f.rec + (size_t)b * f.s_rec;
    t.grid = f.grid + (size_t)b * f.s_grid;
    t.nnodes = f.info[b].x;
    t.grid_depth = f.info[b].y;
    t.simple = 0;
    t.mask = nullptr;
    t.ray_G = 0;
    t.ray_cell = nullptr, t.ray_tri = nullptr, t.ray_more = nullptr, t.ray_excl = nullptr;
    t.ray_r2lo = t.ray_r2hi = 0.0;
    return t;
}
// every query point in every tree (blockIdx.y), eight lanes per query (search_device.hpp: group_search)
struct ForestQueryPayload {
    int id0, id1, id2;
    double wa, wb, wc;
    __device__ __forceinline__ void compute(const TriRec &r, const V3 &, const V3 &mp) {
        id0 = r.id[0], id1 = r.id[1], id2 = r.id[2];
        area_weights(rec_v0(r), rec_v1(r), rec_v2(r), mp, wa, wb, wc);  // calc_barycentric_weights projects the query first (R/triangle.cpp:130)
    }
};
template <int G>
__global__ __launch_bounds__(256) void k_query_forest(ForestDev f, const double *__restrict__ q, int N, int *__restrict__ vid, double *__restrict__ w, size_t comp, int *status) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const DevTree T = forest_tree_dev(f, b);
    constexpr int per_block = 256 / G;
    for (int base = blockIdx.x * per_block; base < N; base += gridDim.x * per_block) {
        const int i = base + threadIdx.x / G;
        const bool valid = i < N;
        const V3 p = valid ? mk(q[i], q[N + i], q[2 * (size_t)N + i]) : mk(0.0, 0.0, 0.0);
        ForestQueryPayload out;
        bool owner;
        const int t = group_search<G>(T, valid, p, lane, out, owner);
        if (!owner) continue;
        const size_t at = (size_t)b * N + i;
        if (t < 0) {
            raise_status(status, t);
            vid[at] = vid[comp + at] = vid[2 * comp + at] = -1;
            w[at] = w[comp + at] = w[2 * comp + at] = 0.0;
            continue;
        }
        vid[at] = out.id0, vid[comp + at] = out.id1, vid[2 * comp + at] = out.id2;
        w[at] = out.wa, w[comp + at] = out.wb, w[2 * comp + at] = out.wc;
    }
}
int launch_query_forest(msm_ctx *ctx, const ForestDev &f, int B, const double *d_q, int N, int *d_vid, double *d_w, size_t comp) {
    if (N <= 0 || B <= 0) return MSM_OK;
    if (query_lanes((long long)N * B) == 4)
        hipLaunchKernelGGL(k_query_forest<4>, dim3((unsigned)std::min((N + 63) / 64, 8192), (unsigned)B), dim3(256), 0, ctx->stream, f, d_q, N, d_vid, d_w, comp, ctx->d_status);
    else
        hipLaunchKernelGGL(k_query_forest<8>, dim3((unsigned)std::min((N + 31) / 32, 8192), (unsigned)B), dim3(256), 0, ctx->stream, f, d_q, N, d_vid, d_w, comp, ctx->d_status);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

// blockIdx.y = subject a; eight lanes per (v, b > a), v-major as in the list
struct PairVertexPayload {  // Octree::get_closest_vertex_ID, R/octree.cpp:216-233
    int best;
    __device__ __forceinline__ void compute(const TriRec &r, const V3 &q, const V3 &) {
        double dist = DBL_MAX;
        const V3 vv[3] = {rec_v0(r), rec_v1(r), rec_v2(r)};
        const int ids[3] = {r.id[0], r.id[1], r.id[2]};
        best = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double d = norm(sub(q, vv[k]));
            if (d < dist) {
                best = ids[k];
                dist = d;
            }
        }
    }
};
template <int G>
__global__ __launch_bounds__(256) void k_group_pairs(ForestDev f, const double *__restrict__ cp, int S, int N, int *__restrict__ pairs, int *status) {
    const int a = blockIdx.y, nb = S - 1 - a, lane = threadIdx.x & 63;
    const long long i = (long long)blockIdx.x * (256 / G) + threadIdx.x / G;
    const bool valid = i < (long long)N * nb;
    if (!__syncthreads_or(valid)) return;
    const int v = valid ? (int)(i / nb) : 0, b = valid ? a + 1 + (int)(i - (long long)v * nb) : a;
    // pairs before subject a: N * sum_{a' < a} (S - 1 - a')
    const long long base = (long long)N * ((long long)a * (S - 1) - (long long)a * (a - 1) / 2);
    const long long p = base + i;
    const size_t comp = (size_t)S * N;
    const V3 q = mk(cp[(size_t)a * N + v], cp[comp + (size_t)a * N + v], cp[2 * comp + (size_t)a * N + v]);
    const DevTree T = forest_tree_dev(f, b);  // differs between the groups of a wavefront: fine, the group exchanges carry no tree state
    PairVertexPayload out;
    out.best = 0;
    bool owner;
    const int t = group_search<G>(T, valid, q, lane, out, owner);
    if (!owner) return;
    if (t < 0) raise_status(status, t);
    pairs[2 * p] = a * N + v;
    pairs[2 * p + 1] = b * N + (t < 0 ? t : out.best);
}
int launch_group_pairs(msm_ctx *ctx, const ForestDev &f, const double *d_cp, int S, int N, int *d_pairs) {
    if (S < 2) return MSM_OK;
    const long long most = (long long)N * (S - 1);
    if (query_lanes(most * (S - 1) / 2) == 4)
        hipLaunchKernelGGL(k_group_pairs<4>, dim3((unsigned)((most + 63) / 64), (unsigned)(S - 1)), dim3(256), 0, ctx->stream, f, d_cp, S, N, d_pairs, ctx->d_status);
    else
        hipLaunchKernelGGL(k_group_pairs<8>, dim3((unsigned)((most + 31) / 32), (unsigned)(S - 1)), dim3(256), 0, ctx->stream, f, d_cp, S, N, d_pairs, ctx->d_status);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

__global__ __launch_bounds__(256) void k_group_moved(const double *__restrict__ rot, int nodes, const double *__restrict__ labels, int L, double *__restrict__ moved) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nodes * L) return;
    const size_t node = i / L;
    const int l = (int)(i - node * L);
    const V3 m = rotate(rot + 9 * node, mk(labels[l], labels[L + l], labels[2 * (size_t)L + l]));
    moved[3 * i] = m.x, moved[3 * i + 1] = m.y, moved[3 * i + 2] = m.z;
}
int launch_group_moved(msm_ctx *ctx, const double *d_rot, int nodes, const double *d_labels, int L, double *d_moved) {
    const size_t n = (size_t)nodes * L;
    if (n == 0) return MSM_OK;
    hipLaunchKernelGGL(k_group_moved, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_rot, nodes, d_labels, L, d_moved);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

__global__ __launch_bounds__(256) void k_group_centres(const double *__restrict__ moved, const double *__restrict__ spacing, int N, int L, double *__restrict__ centres,
                                                        double *__restrict__ sep) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x, M = N * L;
    if (k >= M) return;
    centres[k] = moved[3 * (size_t)k];
    centres[M + k] = moved[3 * (size_t)k + 1];
    centres[2 * (size_t)M + k] = moved[3 * (size_t)k + 2];
    sep[k] = spacing[k / L];
}
int launch_group_centres(msm_ctx *ctx, const double *d_moved_subject, const double *d_spacing_subject, int N, int L, double *d_centres, double *d_sep) {
    const int M = N * L;
    if (M <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_group_centres, dim3((M + 255) / 256), dim3(256), 0, ctx->stream, d_moved_subject, d_spacing_subject, N, L, d_centres, d_sep);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_group_pairwise(msm_ctx *ctx, const GroupArgs &a, const int *qp, const int *qa, const int *qb, int n, double *out) {
    if (n <= 0) return MSM_OK;
    if (a.simmeasure == 4 || a.simmeasure == 5) {
        const size_t lds = sizeof(double) * 4 * 2 * (size_t)std::max(a.patch_cap, 1);  // <= 128 KB: patches hold at most 2048 entries
        if (lds > 160 * 1024) return fail(MSM_ERR_CAPACITY, "group patch of %d entries does not fit in LDS", a.patch_cap);
        if (lds > 64 * 1024) MSM_HIP(hipFuncSetAttribute((const void *)k_group_pairwise<true, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((k_group_pairwise<true, 64>), dim3((n + 3) / 4), dim3(256), lds, ctx->stream, a, qp, qa, qb, n, out);
    } else {
        // half a wavefront per query: two queries share a wavefront's latency; a quarter when the group's patches are small enough (a.pair_lanes)
        if (a.pair_lanes == 16) hipLaunchKernelGGL((k_group_pairwise<false, 16>), dim3((n + 15) / 16), dim3(256), 0, ctx->stream, a, qp, qa, qb, n, out);
        else hipLaunchKernelGGL((k_group_pairwise<false, 32>), dim3((n + 7) / 8), dim3(256), 0, ctx->stream, a, qp, qa, qb, n, out);
    }
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

__global__ __launch_bounds__(256) void k_group_kept(const int *__restrict__ order, int base, const double *__restrict__ kept, int n, double *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int at = order[i] - base;
    out[4 * (size_t)at + 3] = kept[at];
}

// The values of every patch entry beside its id (GroupArgs::pval): a wavefront per (control point, label) row of a subject, blockIdx.y = subject
__global__ __launch_bounds__(256) void k_group_patch_values(GroupArgs a, double *const *__restrict__ pval, const int *__restrict__ node_flags) {
    const int s = blockIdx.y, rows = a.N * a.L;
    const GlobalPtr<int> pp = as_global(a.pptr[s]), pi = as_global(a.pidx[s]);
    double *__restrict__ out = pval[s];
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        if (node_flags && !node_flags[(size_t)s * a.N + row / a.L]) continue;  // (wavefront-uniform: a row per wavefront)
        const int beg = pp[row], end = pp[row + 1], l = row % a.L;
        const GlobalPtr<double> F = as_global(a.F[(size_t)s * a.L + l]);
        for (int e = beg + lane; e < end; e += 64) {
            const int id = pi[e];
            for (int d = 0; d < a.D; ++d) out[(size_t)a.D * e + d] = F[(size_t)d * a.Vt + id];
        }
    }
}

__global__ __launch_bounds__(256) void k_group_patch_dir(GroupArgs a, double *const *__restrict__ pval, GroupPatchRef *__restrict__ dir) {
    const size_t rows = (size_t)a.N * a.L, i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * a.S) return;
    const int s = (int)(i / rows), row = (int)(i - (size_t)s * rows);
    const int beg = a.pptr[s][row], end = a.pptr[s][row + 1];
    GroupPatchRef r;
    r.ids = a.pidx[s] + beg;
    r.vals = pval ? pval[s] + (size_t)a.D * beg : nullptr;
    r.begin = beg, r.count = end - beg;
    r.pad[0] = r.pad[1] = 0;
    dir[i] = r;
}

int launch_group_patch_dir(msm_ctx *ctx, const GroupArgs &a, double *const *pval, GroupPatchRef *dir) {
    const size_t n = (size_t)a.S * a.N * a.L;
    if (n == 0) return MSM_OK;
    hipLaunchKernelGGL(k_group_patch_dir, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, pval, dir);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

__global__ __launch_bounds__(256) void k_group_expand_order(const int *__restrict__ order, const int2 *__restrict__ pairs, int n, int4 *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int p = order[i];
    const int2 ab = pairs[p];
    out[i] = make_int4(p, ab.x, ab.y, 0);
}

int launch_group_expand_order(msm_ctx *ctx, const int *order, const int *pairs, int n, int4 *out) {
    if (n <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_group_expand_order, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, order, reinterpret_cast<const int2 *>(pairs), n, out);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

__global__ __launch_bounds__(256) void k_group_mark_nodes(const int2 *__restrict__ pairs, long long n, int *__restrict__ node_flags) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int2 ab = pairs[i];
    node_flags[ab.x] = 1;
    node_flags[ab.y] = 1;
}

int launch_group_mark_nodes(msm_ctx *ctx, const int *pairs, long long pair0, long long pair1, int *node_flags) {
    const long long n = pair1 - pair0;
    if (n <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_group_mark_nodes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, reinterpret_cast<const int2 *>(pairs) + pair0, n, node_flags);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_group_patch_values(msm_ctx *ctx, const GroupArgs &a, double *const *pval, const int *node_flags) {
    const int rows = a.N * a.L;
    if (rows <= 0 || a.S <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_group_patch_values, dim3((unsigned)std::min((rows + 3) / 4, 4096), (unsigned)a.S), dim3(256), 0, ctx->stream, a, pval, node_flags);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_group_kept(msm_ctx *ctx, const int *order, int base, const double *kept, int n, double *out) {
    if (n <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_group_kept, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, order, base, kept, n, out);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

// the pair list in another order: out[i] = in[order[i]] (msm_group_set_pair_layout)
__global__ __launch_bounds__(256) void k_group_permute_pairs(const int2 *__restrict__ in, const int *__restrict__ order, int n, int2 *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = in[order[i]];
}

int launch_group_permute_pairs(msm_ctx *ctx, const int *d_in, const int *d_order, int n, int *d_out) {
    if (n <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_group_permute_pairs, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, reinterpret_cast<const int2 *>(d_in), d_order, n,
                       reinterpret_cast<int2 *>(d_out));
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

int launch_group_triplet(msm_ctx *ctx, const GroupArgs &a, const int *qt, const int *qa, const int *qb, const int *qc, int n, double *out) {
    if (n <= 0) return MSM_OK;
    hipLaunchKernelGGL(k_group_triplet, dim3((n + 127) / 128), dim3(128), 0, ctx->stream, a, qt, qa, qb, qc, n, out);
    MSM_HIP(hipGetLastError());
    return MSM_OK;
}

}  // namespace msm
