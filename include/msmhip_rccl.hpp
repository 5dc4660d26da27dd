// msmhip_rccl.hpp -- the collectives of a groupwise (gMSM) registration for a C++ host: one process per GPU, RCCL over xGMI.
// Header only; link with -lrccl (and libmsmhip).  The Python side of this repo does the same through torch.distributed
// (newmsm_amd/dist.py); the ABI underneath is identical: libmsmhip fills and reads plain device buffers, the caller moves them.
//
//   bootstrap       ncclGetUniqueId on rank 0, handed to the other ranks by the launcher's means (MPI, a file, an environment
//                   variable: write_unique_id / read_unique_id below are the file variant for single-node launchers)
//   set-up          sharded_group_setup: every rank runs get_patch_data (M/DiscreteGroupModel.cpp:88-121) for ITS subjects, then
//                   three all-gathers (resampled feature maps, patch row pointers, patch index lists; shards padded to the
//                   largest) give every rank every subject: msm_group_export_subject_dev -> ncclAllGather -> _import_subject_dev
//   template        group_template_update: what gMSM_scripts/run_gMSM.sh:66-139 does with files and wb_command -- the mean of the
//                   registered spheres (renormalised to the radius) and mean / variance of the features -- as one all-reduce
//   label steps     no collective on one node (the ranks deliver their slices through shared pinned host memory, msm_host_register +
//                   msm_group_fusion_move_dev: INTEGRATION.md section 3b); gather_label_step is the multi-node route
#pragma once

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "msmhip.h"

namespace msmhip {
namespace rccl {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
inline void check_hip(hipError_t e, const char *what) {
    if (e != hipSuccess) throw Error(std::string(what) + ": " + hipGetErrorString(e));
}
inline void check_nccl(ncclResult_t r, const char *what) {
    if (r != ncclSuccess) throw Error(std::string(what) + ": " + ncclGetErrorString(r));
}
inline void check_msm(int st, const char *what) {
    if (st != MSM_OK) throw Error(std::string(what) + ": " + msm_last_error());
}

// contiguous, balanced shard of range(n) for `rank` (the same rule as newmsm_amd/dist.py: shard)
inline void shard(int n, int rank, int world, int &lo, int &hi) {
    const int base = n / world, extra = n % world;
    lo = rank * base + std::min(rank, extra);
    hi = lo + base + (rank < extra ? 1 : 0);
}

inline void write_unique_id(const ncclUniqueId &id, const std::string &path) {
    const std::string tmp = path + ".tmp";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) throw Error("cannot write " + tmp);
    std::fclose(f);
    if (std::rename(tmp.c_str(), path.c_str()) != 0) throw Error("cannot publish " + path);
}
inline bool read_unique_id(ncclUniqueId &id, const std::string &path) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    const bool ok = std::fread(&id, sizeof(id), 1, f) == 1;
    std::fclose(f);
    return ok;
}

// one rank's end of the communicator; the stream is the msm_ctx's (msm_ctx_stream), so libmsmhip's copies and the collectives
// are ordered without further synchronisation
class Comm {
public:
    Comm(int rank, int world, const ncclUniqueId &id, hipStream_t stream) : rank_(rank), world_(world), stream_(stream) {
        check_nccl(ncclCommInitRank(&comm_, world, id, rank), "ncclCommInitRank");
    }
    ~Comm() {
        if (comm_) (void)ncclCommDestroy(comm_);
    }
    Comm(const Comm &) = delete;
    Comm &operator=(const Comm &) = delete;
    int rank() const { return rank_; }
    int world() const { return world_; }
    hipStream_t stream() const { return stream_; }
    ncclComm_t raw() const { return comm_; }

    template <typename T>
    void all_gather(const T *send, T *recv, size_t count_per_rank) const {
        check_nccl(ncclAllGather(send, recv, count_per_rank * sizeof(T), ncclChar, comm_, stream_), "ncclAllGather");
    }
    void all_reduce_sum(double *buf, size_t n) const { check_nccl(ncclAllReduce(buf, buf, n, ncclDouble, ncclSum, comm_, stream_), "ncclAllReduce"); }
    void gather_to(const double *send, double *recv /* world x count on dst */, size_t count, int dst) const {
        // RCCL has no gather: grouped point-to-point, as its documentation prescribes
        check_nccl(ncclGroupStart(), "ncclGroupStart");
        check_nccl(ncclSend(send, count, ncclDouble, dst, comm_, stream_), "ncclSend");
        if (rank_ == dst)
            for (int r = 0; r < world_; ++r) check_nccl(ncclRecv(recv + (size_t)r * count, count, ncclDouble, r, comm_, stream_), "ncclRecv");
        check_nccl(ncclGroupEnd(), "ncclGroupEnd");
    }
    void synchronize() const { check_hip(hipStreamSynchronize(stream_), "hipStreamSynchronize"); }

private:
    int rank_, world_;
    hipStream_t stream_;
    ncclComm_t comm_ = nullptr;
};

namespace detail {
template <typename T>
struct DeviceArray {
    T *p = nullptr;
    size_t n = 0;
    explicit DeviceArray(size_t count) : n(count) { check_hip(hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T)), "hipMalloc"); }
    ~DeviceArray() {
        if (p) (void)hipFree(p);
    }
    DeviceArray(const DeviceArray &) = delete;
    DeviceArray &operator=(const DeviceArray &) = delete;
};
// small host buffers the copy engine reads and writes: page-locked memory of their own (an asynchronous copy into a pageable std::vector makes the HIP
// runtime page-lock whole pages of the heap behind the caller's back -- shared with whatever else lives there; csrc/stager.cpp has the story)
template <typename T>
struct PinnedArray {
    T *p = nullptr;
    size_t n = 0;
    explicit PinnedArray(size_t count) : n(count) {
        check_hip(hipHostMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T)), "hipHostMalloc");
        std::fill(p, p + count, T(0));
    }
    ~PinnedArray() {
        if (p) (void)hipHostFree(p);
    }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    PinnedArray(const PinnedArray &) = delete;
    PinnedArray &operator=(const PinnedArray &) = delete;
};
}  // namespace detail

// Groupwise set-up with the subjects sharded over the ranks.  Every rank has created the group with ALL control grids and labels
// and with the data (msm_group_set_subject) of at least its own subjects.  Returns this rank's subjects.
inline std::vector<int32_t> sharded_group_setup(msm_group *g, const Comm &c) {
    int32_t S = 0, N = 0, L = 0, D = 0, Vt = 0;
    check_msm(msm_group_dims(g, &S, &N, &L, &D, &Vt), "msm_group_dims");
    int lo, hi;
    shard(S, c.rank(), c.world(), lo, hi);
    std::vector<int32_t> mine;
    for (int s = lo; s < hi; ++s) mine.push_back(s);
    // the pair list control point by control point, so that the contiguous slice of it a rank evaluates in every label step is a region of the sphere
    // (msmhip.h: msm_group_set_pair_layout; newmsm_amd/dist.py does the same) -- for ANY number of ranks, one included: the optimiser consumes pairs and
    // costs in list order, and a launched run must not depend on how many ranks it was given
    check_msm(msm_group_set_pair_layout(g, 1), "msm_group_set_pair_layout");
    check_msm(msm_group_setup_subjects(g, mine.data(), (int32_t)mine.size()), "msm_group_setup_subjects");
    if (c.world() > 1) {
        int nmax = 0;
        for (int r = 0; r < c.world(); ++r) {
            int a, b;
            shard(S, r, c.world(), a, b);
            nmax = std::max(nmax, b - a);
        }
        // 1. how long every subject's index list is
        detail::PinnedArray<long long> counts(nmax), all_counts((size_t)c.world() * nmax);
        for (size_t k = 0; k < mine.size(); ++k) {
            int64_t n = 0;
            check_msm(msm_group_export_subject_dev(g, mine[k], nullptr, nullptr, nullptr, 0, &n), "msm_group_export_subject_dev (count)");
            counts[k] = n;
        }
        {
            detail::DeviceArray<long long> d_counts(nmax), d_all((size_t)c.world() * nmax);
            check_hip(hipMemcpyAsync(d_counts.p, counts.p, sizeof(long long) * nmax, hipMemcpyHostToDevice, c.stream()), "copy counts");
            c.all_gather(d_counts.p, d_all.p, (size_t)nmax);
            check_hip(hipMemcpyAsync(all_counts.p, d_all.p, sizeof(long long) * all_counts.n, hipMemcpyDeviceToHost, c.stream()), "copy counts back");
            c.synchronize();
        }
        const size_t imax = (size_t)std::max<long long>(1, *std::max_element(all_counts.p, all_counts.p + all_counts.n));
        const size_t per_F = (size_t)L * D * Vt, per_pp = (size_t)N * L + 1;
        // 2. this rank's shard in device buffers libmsmhip fills, 3. three all-gathers, 4. the other ranks' subjects read back from them
        detail::DeviceArray<double> F(nmax * per_F), aF((size_t)c.world() * nmax * per_F);
        detail::DeviceArray<int32_t> pp(nmax * per_pp), app((size_t)c.world() * nmax * per_pp), pi(nmax * imax), api((size_t)c.world() * nmax * imax);
        check_hip(hipMemsetAsync(pi.p, 0, sizeof(int32_t) * nmax * imax, c.stream()), "hipMemsetAsync");
        // the stream contract of the ..._dev entry points (msmhip.h): the library writes these buffers on ITS stream, which does not order against ours --
        // it waits for the fill above (an event, no host wait); a fill landing after the export would wipe exported rows
        msm_ctx *lib = msm_group_context(g);
        check_msm(msm_ctx_wait_stream(lib, c.stream()), "msm_ctx_wait_stream");
        for (size_t k = 0; k < mine.size(); ++k)
            check_msm(msm_group_export_subject_dev(g, mine[k], F.p + k * per_F, pp.p + k * per_pp, pi.p + k * imax, (int64_t)imax, nullptr), "msm_group_export_subject_dev");
        c.all_gather(F.p, aF.p, nmax * per_F);
        c.all_gather(pp.p, app.p, nmax * per_pp);
        c.all_gather(pi.p, api.p, nmax * imax);
        check_msm(msm_ctx_wait_stream(lib, c.stream()), "msm_ctx_wait_stream");  // the imports below read what the collectives are still writing
        for (int r = 0; r < c.world(); ++r) {
            if (r == c.rank()) continue;
            int a, b;
            shard(S, r, c.world(), a, b);
            if (b <= a) continue;
            // rank r's shard in one call: its slots are consecutive in the gathered buffers (one range check launch, one synchronisation)
            std::vector<int32_t> theirs;
            std::vector<int64_t> np;
            for (int s = a; s < b; ++s) {
                theirs.push_back(s);
                np.push_back((int64_t)all_counts[(size_t)r * nmax + (s - a)]);
            }
            const size_t k = (size_t)r * nmax;
            check_msm(msm_group_import_subjects_dev(g, theirs.data(), (int32_t)theirs.size(), aF.p + k * per_F, (int64_t)per_F, app.p + k * per_pp, (int64_t)per_pp,
                                                    api.p + k * imax, (int64_t)imax, np.data()),
                      "msm_group_import_subjects_dev");
        }
    }
    check_msm(msm_group_finalize(g), "msm_group_finalize");
    return mine;
}

// The group-mean template update.  local_xyz: n_local registered spheres (n_local x V x 3, host); features (optional): n_local x D x V.
// Returns the renormalised mean sphere (V x 3) and, if features were given, their mean and variance over all subjects (D x V each).
struct TemplateUpdate {
    std::vector<double> sphere, mean, variance;
    long long n_subjects = 0;
};
inline TemplateUpdate group_template_update(const double *local_xyz, int n_local, int V, const double *local_feat, int D, const Comm &c, double radius = 100.0) {
    const size_t nx = (size_t)3 * V, nf = local_feat ? (size_t)D * V : 0, total = nx + 2 * nf + 1;
    std::vector<double> acc(total, 0.0);
    for (int s = 0; s < n_local; ++s) {
        for (size_t i = 0; i < nx; ++i) acc[i] += local_xyz[(size_t)s * nx + i];
        for (size_t i = 0; i < nf; ++i) {
            const double f = local_feat[(size_t)s * nf + i];
            acc[nx + i] += f;
            acc[nx + nf + i] += f * f;
        }
    }
    acc[total - 1] = (double)n_local;
    detail::DeviceArray<double> d(total);
    detail::PinnedArray<double> pinned(total);  // the copy engine's side of the two transfers (see PinnedArray)
    std::copy(acc.begin(), acc.end(), pinned.p);
    check_hip(hipMemcpyAsync(d.p, pinned.p, sizeof(double) * total, hipMemcpyHostToDevice, c.stream()), "upload accumulators");
    c.all_reduce_sum(d.p, total);
    check_hip(hipMemcpyAsync(pinned.p, d.p, sizeof(double) * total, hipMemcpyDeviceToHost, c.stream()), "download accumulators");
    c.synchronize();
    std::copy(pinned.p, pinned.p + total, acc.begin());
    TemplateUpdate out;
    out.n_subjects = (long long)std::llround(acc[total - 1]);
    const double n = std::max(1.0, acc[total - 1]);
    out.sphere.resize(nx);
    for (int v = 0; v < V; ++v) {
        const double x = acc[3 * (size_t)v] / n, y = acc[3 * (size_t)v + 1] / n, z = acc[3 * (size_t)v + 2] / n;
        const double len = std::sqrt(x * x + y * y + z * z), k = len > 0 ? radius / len : 0.0;
        out.sphere[3 * (size_t)v] = x * k, out.sphere[3 * (size_t)v + 1] = y * k, out.sphere[3 * (size_t)v + 2] = z * k;
    }
    if (nf) {
        out.mean.resize(nf);
        out.variance.resize(nf);
        for (size_t i = 0; i < nf; ++i) {
            out.mean[i] = acc[nx + i] / n;
            out.variance[i] = std::max(0.0, acc[nx + nf + i] / n - out.mean[i] * out.mean[i]);
        }
    }
    return out;
}

// A label step with the cliques sharded over ranks on SEVERAL nodes: this rank's slice into `send` (device, 4 * pmax + 8 * tmax doubles,
// slices padded to the largest), gathered on `dst` into recv (device, world x that).  On one node use shared pinned host memory instead.
inline void gather_label_step(msm_group *g, const int32_t *labeling, int32_t label, double *send, double *recv, int dst, const Comm &c) {
    int32_t nodes = 0, P = 0, T = 0;
    check_msm(msm_group_sizes(g, &nodes, &P, &T), "msm_group_sizes");
    int p0, p1, t0, t1, pmax = 0, tmax = 0;
    for (int r = 0; r < c.world(); ++r) {
        int a, b;
        shard(P, r, c.world(), a, b);
        pmax = std::max(pmax, b - a);
        shard(T, r, c.world(), a, b);
        tmax = std::max(tmax, b - a);
    }
    shard(P, c.rank(), c.world(), p0, p1);
    shard(T, c.rank(), c.world(), t0, t1);
    check_msm(msm_group_fusion_move_dev(g, labeling, label, p0, p1, t0, t1, send, send + 4 * (size_t)pmax), "msm_group_fusion_move_dev");
    c.gather_to(send, recv, 4 * (size_t)pmax + 8 * (size_t)tmax, dst);
    c.synchronize();
}

}  // namespace rccl
}  // namespace msmhip
