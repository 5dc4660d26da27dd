"""Thin Python mirror of the reference's C++ interfaces on top of the libmsmhip C ABI.

Names follow newMSM: Mesh / Octree queries and the free functions of resampler.h
(/root/reference/libraries/msm-newresampler/src/resampler.h:38-53), and the evaluator interface of
DiscreteCostFunction (/root/reference/libraries/msm-newmeshreg/src/DiscreteCostFunction.h:41-59,
:161-209).  Points are (N,3) numpy arrays here; the ABI's 3 x N SoA layout is produced at the boundary.
This module is plumbing for tests and bench.py -- all computation happens in libmsmhip.so.
"""
import ctypes as C

import numpy as np

from ._lib import CostParams, MsmError, c_dp, c_ip, c_lp, check, lib

WEIGHTS_PROJECTED = 0
WEIGHTS_RAW = 1
KINDS = dict(univariate=0, multivariate=1, patchwise=2, ho_univariate=3, ho_multivariate=4)


def _soa(points):
    """(N,3) -> contiguous 3 x N"""
    a = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(-1, 3).T)
    return a, a.ctypes.data_as(c_dp)


def _aos(soa, n):
    return np.ascontiguousarray(np.asarray(soa).reshape(3, n).T)


def _tri_soa(tri):
    a = np.ascontiguousarray(np.asarray(tri, dtype=np.int32).reshape(-1, 3).T)
    return a, a.ctypes.data_as(c_ip)


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(c_ip)


# ------------------------------------------------------------------ host helpers
def device_count():
    return lib().msm_device_count()


def icosphere_counts(order):
    v, t = C.c_int32(), C.c_int32()
    check(lib().msm_icosphere_counts(order, C.byref(v), C.byref(t)))
    return v.value, t.value


def make_mesh_from_icosa(order, radius=100.0):
    """make_mesh_from_icosa(order) + true_rescale(radius). Returns (xyz[V,3], tri[T,3])."""
    V, T = icosphere_counts(order)
    xyz = np.zeros((3, V))
    tri = np.zeros((3, T), dtype=np.int32)
    check(lib().msm_icosphere(order, float(radius if radius else 0.0), xyz.ctypes.data_as(c_dp), tri.ctypes.data_as(c_ip)))
    return np.ascontiguousarray(xyz.T), np.ascontiguousarray(tri.T)


def mesh_adjacency(tri, V):
    t, pt = _tri_soa(tri)
    T = t.shape[1]
    nbr_ptr = np.zeros(V + 1, dtype=np.int32)
    tid_ptr = np.zeros(V + 1, dtype=np.int32)
    check(lib().msm_mesh_adjacency(pt, V, T, nbr_ptr.ctypes.data_as(c_ip), None, tid_ptr.ctypes.data_as(c_ip), None))
    nbr = np.zeros(nbr_ptr[-1], dtype=np.int32)
    tid = np.zeros(tid_ptr[-1], dtype=np.int32)
    check(lib().msm_mesh_adjacency(pt, V, T, nbr_ptr.ctypes.data_as(c_ip), nbr.ctypes.data_as(c_ip), tid_ptr.ctypes.data_as(c_ip),
                                   tid.ctypes.data_as(c_ip)))
    return nbr_ptr, nbr, tid_ptr, tid


def vertex_areas(xyz, tri):
    x, px = _soa(xyz)
    t, pt = _tri_soa(tri)
    out = np.zeros(x.shape[1])
    check(lib().msm_vertex_areas(px, pt, x.shape[1], t.shape[1], out.ctypes.data_as(c_dp)))
    return out


def cp_spacings(xyz, tri):
    x, px = _soa(xyz)
    t, pt = _tri_soa(tri)
    ms = np.zeros(x.shape[1])
    mvd = C.c_double()
    check(lib().msm_cp_spacings(px, pt, x.shape[1], t.shape[1], ms.ctypes.data_as(c_dp), C.byref(mvd)))
    return ms, mvd.value


def label_sampling_grid(sg_order, max_dist, abs_is_int=False, cap=4096):
    s = np.zeros((3, cap))
    b = np.zeros((3, cap))
    ns, nb = C.c_int32(), C.c_int32()
    check(lib().msm_label_sampling_grid(sg_order, float(max_dist), int(abs_is_int), cap, s.ctypes.data_as(c_dp), C.byref(ns),
                                        b.ctypes.data_as(c_dp), C.byref(nb)))
    return np.ascontiguousarray(s[:, : ns.value].T), np.ascontiguousarray(b[:, : nb.value].T)


def rescale_sampling_grid(samples, scale):
    s, ps = _soa(samples)
    n = s.shape[1]
    out = np.zeros((3, n))
    sc = C.c_double(scale)
    check(lib().msm_rescale_sampling_grid(ps, n, C.byref(sc), out.ctypes.data_as(c_dp)))
    return np.ascontiguousarray(out.T), sc.value


def estimate_rotation_matrix(ci, index):
    R = np.zeros(9)
    check(lib().msm_rotation_matrix(_d(ci)[1], _d(index)[1], R.ctypes.data_as(c_dp)))
    return R.reshape(3, 3)


def cp_rotations(centre, cp_xyz):
    x, px = _soa(cp_xyz)
    rot = np.zeros((x.shape[1], 9))
    check(lib().msm_cp_rotations(_d(centre)[1], px, x.shape[1], rot.ctypes.data_as(c_dp)))
    return rot


def octree_signature(xyz, tri):
    """[host] statistics and leaf signature of the search tree of a mesh (testing hook)."""
    x, px = _soa(xyz)
    t, pt = _tri_soa(tri)
    stats = (C.c_int64 * 5)()
    sig = C.c_uint64(0)
    check(lib().msm_octree_signature(px, pt, x.shape[1], t.shape[1], stats, C.byref(sig)))
    return dict(nodes=stats[0], leaves=stats[1], depth=stats[2], refs=stats[3], max_leaf=stats[4]), sig.value


def ray_table_check(xyz, tri, nsamples=20000, seed=1):
    """[host] the direction table's guarantee checked point by point against the reference's leaf search (testing hook, msm_ray_table_check)."""
    x, px = _soa(xyz)
    t, pt = _tri_soa(tri)
    rep = (C.c_int64 * 10)()
    check(lib().msm_ray_table_check(px, pt, x.shape[1], t.shape[1], nsamples, seed, rep))
    return dict(points=rep[0], by_float=rep[1], by_fp64=rep[2], open=rep[3], violations=rep[4], unusable=rep[5], with_exclusions=rep[6], simple=bool(rep[7]), boxes_checked=rep[8], refused_by_boxes=rep[9])


def estimate_triplets(tri):
    t, pt = _tri_soa(tri)
    out = np.zeros((t.shape[1], 3), dtype=np.int32)
    check(lib().msm_estimate_triplets(pt, t.shape[1], out.ctypes.data_as(c_ip)))
    return out


def estimate_pairs(tri, V):
    t, pt = _tri_soa(tri)
    n = lib().msm_estimate_pairs(pt, V, t.shape[1], None)
    if n < 0:
        check(n)
    out = np.zeros((n, 2), dtype=np.int32)
    lib().msm_estimate_pairs(pt, V, t.shape[1], out.ctypes.data_as(c_ip))
    return out


# ------------------------------------------------------------------ context / mesh
class Context:
    """One per GPU (msm_ctx)."""

    def __init__(self, device=0, stream=None):
        L = lib()
        self.h = L.msm_ctx_create(device) if stream is None else L.msm_ctx_create_on_stream(device, C.c_void_p(stream))
        if not self.h:
            raise MsmError(-5, L.msm_last_error().decode())
        self.device = device

    def synchronize(self):
        check(lib().msm_ctx_synchronize(self.h))

    @property
    def stream(self):
        return lib().msm_ctx_stream(self.h)

    def wait_stream(self, hip_stream=None):
        """everything the library queues on this context from now on waits for what `hip_stream` (a hipStream_t value; None = the default stream)
        holds now (msm_ctx_wait_stream): the ordering a caller owes the ..._dev entry points for buffers its own stream is still filling"""
        check(lib().msm_ctx_wait_stream(self.h, C.c_void_p(hip_stream or 0)))

    def staging_stats(self):
        """pinned staging blocks of this context: dict(blocks, bytes, allocated, waits) (msm_ctx_staging_stats)"""
        out = (C.c_int64 * 4)()
        check(lib().msm_ctx_staging_stats(self.h, out))
        return dict(blocks=int(out[0]), bytes=int(out[1]), allocated=int(out[2]), waits=int(out[3]))

    def host_array(self, shape, dtype=np.float64):
        """A numpy array in pinned host memory mapped into the GPU's address space (msm_host_alloc): passed as an output array
        the kernels write it directly.  Lives as long as the context."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = lib().msm_host_alloc(self.h, max(n, 8))
        if not p:
            raise MsmError(-2, lib().msm_last_error().decode())
        buf = (C.c_char * max(n, 8)).from_address(p)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def scratch_host_array(self, name, shape, dtype=np.float64):
        """a view of a grow-only pinned buffer kept per `name` on this context.  When it has to grow a new block is taken and the old
        one is NOT freed: arrays handed out earlier may still be referenced by the caller (a freed block is unmapped -- reading such a
        view would fault).  Blocks grow by doubling, so what stays behind is less than the final block; all go with the context."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        bufs = self.__dict__.setdefault("_scratch_pinned", {})
        cur = bufs.get(name)
        if cur is None or cur[1] < n:
            cap = max(2 * n, 4096) if cur is not None else max(n + n // 8, 4096)
            p = lib().msm_host_alloc(self.h, cap)
            if not p:
                raise MsmError(-2, lib().msm_last_error().decode())
            cur = bufs[name] = (p, cap)
        buf = (C.c_char * cur[1]).from_address(cur[0])
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def release_host_array(self, arr):
        """gives a Context.host_array block back (msm_host_free); the caller guarantees that no view of it is used afterwards"""
        if getattr(self, "h", None):
            lib().msm_host_free(self.h, C.c_void_p(arr.ctypes.data))

    def time_queries(self, on=True):
        """HIP events around the search kernel of query_triangles / closest_vertex calls (msm_ctx_time_queries)"""
        check(lib().msm_ctx_time_queries(self.h, int(on)))

    def query_kernel_ms(self):
        ms = C.c_double(-1.0)
        check(lib().msm_ctx_query_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    @staticmethod
    def query_lanes(n_queries):
        """lanes per query (4 or 8) of the search kernels for a launch of n_queries (msm_query_lanes): the k_query<G> instantiation that runs"""
        return int(lib().msm_query_lanes(int(n_queries)))

    def forest_signatures(self, xyz_sets, tri):
        """leaf signatures (Mesh.octree_signature) of the trees of B coordinate sets over one triangle list, built together as a forest"""
        sets = [np.ascontiguousarray(np.asarray(x, dtype=np.float64).T) for x in xyz_sets]
        big = np.ascontiguousarray(np.stack(sets))  # B x 3 x V
        t, pt = _tri_soa(tri)
        sig = (C.c_uint64 * len(sets))()
        check(lib().msm_octree_forest_signatures(self.h, big.ctypes.data_as(c_dp), big.shape[2], pt, t.shape[1], len(sets), sig))
        return [int(v) for v in sig]

    def register_host(self, address, nbytes):
        """Pin caller-owned host memory (e.g. a shared-memory mapping) and map it into the GPU's address space
        (msm_host_register): arrays inside it are then written by the GPU directly, like Context.host_array's."""
        check(lib().msm_host_register(self.h, C.c_void_p(address), nbytes))

    def unregister_host(self, address):
        lib().msm_host_free(self.h, C.c_void_p(address))

    def close(self):
        if getattr(self, "h", None):
            lib().msm_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class Mesh:
    """newresampler::Mesh as the hot path sees it + its Octree (built lazily on the device side)."""

    def __init__(self, ctx, xyz, tri):
        self.ctx = ctx
        x, px = _soa(xyz)
        t, pt = _tri_soa(tri)
        self.V, self.T = x.shape[1], t.shape[1]
        self.tri = np.ascontiguousarray(t.T)
        self.h = lib().msm_mesh_create(ctx.h, px, self.V, pt, self.T)
        if not self.h:
            raise MsmError(-1, lib().msm_last_error().decode())

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            lib().msm_mesh_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()

    def set_coords(self, xyz):
        check(lib().msm_mesh_update_coords(self.h, _soa(xyz)[1]))

    def get_coords(self):
        out = np.zeros((3, self.V))
        check(lib().msm_mesh_get_coords(self.h, out.ctypes.data_as(c_dp)))
        return np.ascontiguousarray(out.T)

    def prepare_search(self, wait=True):
        """builds the target-side search structures (msm_mesh_prepare_search); returns True when nothing is pending"""
        ready = C.c_int32()
        check(lib().msm_mesh_prepare_search(self.h, int(bool(wait)), C.byref(ready)))
        return bool(ready.value)

    def unfold(self, radius=100.0):
        """unfold (M/reg_tools.cpp:131-178) on the current coordinates; returns (passes, folded vertices of the first pass)."""
        passes, first = C.c_int32(), C.c_int32()
        check(lib().msm_mesh_unfold(self.h, float(radius), C.byref(passes), C.byref(first)))
        return passes.value, first.value

    def set_pvalues(self, feat):
        f, pf = _d(np.atleast_2d(feat))
        assert f.shape[1] == self.V
        check(lib().msm_mesh_set_features(self.h, pf, f.shape[0]))

    def octree_stats(self):
        s = (C.c_int64 * 5)()
        check(lib().msm_mesh_octree_stats(self.h, s))
        return dict(nodes=s[0], leaves=s[1], depth=s[2], refs=s[3], max_leaf=s[4])

    def octree_signature(self):
        """(stats, leaf signature) of the tree as it sits in HBM -- comparable with octree_signature(xyz, tri) of the host build"""
        s = (C.c_int64 * 5)()
        sig = C.c_uint64()
        check(lib().msm_mesh_octree_signature(self.h, s, C.byref(sig)))
        return dict(nodes=s[0], leaves=s[1], depth=s[2], refs=s[3], max_leaf=s[4]), sig.value

    # Octree::get_closest_triangle + get_barycentric_weights
    def query_triangles_soa(self, q_soa, tri, vid, w, mode=WEIGHTS_PROJECTED):
        """msm_query_triangles on the ABI's own layout, no transposes and no allocation: q_soa 3 x N float64, tri N int32, vid 3 x N int32, w 3 x N
        float64 (C-contiguous).  Arrays from Context.host_array (pinned, mapped) are read and written by the copy engine directly -- the call is the
        upload, the kernel, three downloads and one synchronisation.  Returns the status."""
        N = q_soa.shape[1]
        assert q_soa.flags.c_contiguous and tri.flags.c_contiguous and vid.flags.c_contiguous and w.flags.c_contiguous and q_soa.dtype == np.float64
        assert tri.shape == (N,) and vid.shape == (3, N) and w.shape == (3, N) and tri.dtype == np.int32 and vid.dtype == np.int32 and w.dtype == np.float64
        return lib().msm_query_triangles(self.h, q_soa.ctypes.data_as(c_dp), N, tri.ctypes.data_as(c_ip), vid.ctypes.data_as(c_ip), w.ctypes.data_as(c_dp), mode)

    def query_triangles(self, q, mode=WEIGHTS_PROJECTED, check_status=True):
        x, px = _soa(q)
        N = x.shape[1]
        tri = np.zeros(N, dtype=np.int32)
        vid = np.zeros((3, N), dtype=np.int32)
        w = np.zeros((3, N))
        st = lib().msm_query_triangles(self.h, px, N, tri.ctypes.data_as(c_ip), vid.ctypes.data_as(c_ip), w.ctypes.data_as(c_dp), mode)
        if check_status:
            check(st)
        return st, tri, np.ascontiguousarray(vid.T), np.ascontiguousarray(w.T)

    def get_closest_vertex_ID(self, q):
        x, px = _soa(q)
        out = np.zeros(x.shape[1], dtype=np.int32)
        check(lib().msm_closest_vertex(self.h, px, x.shape[1], out.ctypes.data_as(c_ip)))
        return out


# ------------------------------------------------------------------ resampler free functions
def get_adaptive_barycentric_weights(in_mesh, new_mesh, excl=None):
    nnz = C.c_int64()
    pe = _d(excl)[1] if excl is not None else None
    check(lib().msm_adaptive_barycentric_weights(in_mesh.h, new_mesh.h, pe, None, None, None, 0, C.byref(nnz)))
    rp = np.zeros(new_mesh.V + 1, dtype=np.int32)
    col = np.zeros(nnz.value, dtype=np.int32)
    val = np.zeros(nnz.value)
    check(lib().msm_adaptive_barycentric_weights(in_mesh.h, new_mesh.h, pe, rp.ctypes.data_as(c_ip), col.ctypes.data_as(c_ip),
                                                 val.ctypes.data_as(c_dp), nnz.value, C.byref(nnz)))
    return rp, col, val


def metric_resample(in_mesh, data, new_mesh, excl=None, out=None):
    """metric_resample (R/resampler.cpp:304-309); with excl (the EXCL mesh's values on in_mesh) returns (data, resampled mask).  out (optional): the
    D x V(new) result array -- one from Context.host_array is written by the copy engine directly (no staging memcpy)."""
    d, pd = _d(np.atleast_2d(data))
    if out is None:
        out = np.zeros((d.shape[0], new_mesh.V))
    assert out.shape == (d.shape[0], new_mesh.V) and out.flags.c_contiguous and out.dtype == np.float64
    if excl is None:
        check(lib().msm_metric_resample(in_mesh.h, pd, d.shape[0], new_mesh.h, None, out.ctypes.data_as(c_dp), None))
        return out
    eo = np.zeros(new_mesh.V)
    check(lib().msm_metric_resample(in_mesh.h, pd, d.shape[0], new_mesh.h, _d(excl)[1], out.ctypes.data_as(c_dp), eo.ctypes.data_as(c_dp)))
    return out, eo


def create_exclusion(data, thrl, thru):
    """create_exclusion (R/mesh.cpp:1257-1273) of a D x V matrix"""
    d, pd = _d(np.atleast_2d(data))
    out = np.zeros(d.shape[1])
    check(lib().msm_create_exclusion(pd, d.shape[0], d.shape[1], float(thrl), float(thru), out.ctypes.data_as(c_dp)))
    return out


def sphere_project_warp(sphere, from_mesh, to_xyz):
    s, ps = _soa(sphere)
    s = s.copy()
    check(lib().msm_sphere_project_warp(from_mesh.h, _soa(to_xyz)[1], s.ctypes.data_as(c_dp), s.shape[1]))
    return np.ascontiguousarray(s.T)


def sphere_project_warp_mesh(sphere_mesh, from_mesh, to_xyz):
    """msm_mesh_sphere_project_warp: the coordinates `sphere_mesh` holds moved through (from_mesh -> to_xyz), in place on the device"""
    check(lib().msm_mesh_sphere_project_warp(sphere_mesh.h, from_mesh.h, _soa(to_xyz)[1]))


def resample_anatomy_grid(cp_xyz, cp_tri, levels, rad=100.0):
    """Mesh_registration::resample_anatomy (M/mesh_registration.cpp:250-332) without its surface_resample call [host]: the control grid retessellated
    `levels` (= anatgrid - CPgrid) times and rescaled to `rad` -> dict(sphere_xyz, sphere_tri, w_ptr, w_cp, w_val (_ANATbaryweights as CSR),
    face_ptr, face_idx (NEARESTFACES as CSR, the reference's order)): the arguments of DiscreteCostFunction.set_anatomical besides the two anatomies."""
    x, px = _soa(cp_xyz)
    t, pt = _tri_soa(cp_tri)
    N, Tc = x.shape[1], t.shape[1]
    va, ta = C.c_int32(), C.c_int32()
    check(lib().msm_resample_anatomy_grid(px, N, pt, Tc, int(levels), float(rad), C.byref(va), C.byref(ta), None, None, None, None, None, None, None))
    Va, Ta = va.value, ta.value
    axyz, atri = np.zeros((3, Va)), np.zeros((3, Ta), dtype=np.int32)
    w_ptr, w_cp, w_val = np.zeros(Va + 1, dtype=np.int32), np.zeros(3 * Va, dtype=np.int32), np.zeros(3 * Va)
    face_ptr, face_idx = np.zeros(Tc + 1, dtype=np.int32), np.zeros(Ta, dtype=np.int32)
    check(lib().msm_resample_anatomy_grid(px, N, pt, Tc, int(levels), float(rad), C.byref(va), C.byref(ta), axyz.ctypes.data_as(c_dp), atri.ctypes.data_as(c_ip),
                                          w_ptr.ctypes.data_as(c_ip), w_cp.ctypes.data_as(c_ip), w_val.ctypes.data_as(c_dp), face_ptr.ctypes.data_as(c_ip),
                                          face_idx.ctypes.data_as(c_ip)))
    n = int(w_ptr[-1])
    return dict(sphere_xyz=np.ascontiguousarray(axyz.T), sphere_tri=np.ascontiguousarray(atri.T), w_ptr=w_ptr, w_cp=w_cp[:n].copy(), w_val=w_val[:n].copy(),
                face_ptr=face_ptr, face_idx=face_idx)


def barycentric_coords_resample(from_mesh, coords, q):
    x, px = _soa(q)
    out = np.zeros((3, x.shape[1]))
    check(lib().msm_barycentric_coords_resample(from_mesh.h, _soa(coords)[1], px, x.shape[1], out.ctypes.data_as(c_dp)))
    return np.ascontiguousarray(out.T)


def smooth_data(orig_mesh, data, sph_low, sigma, excl=None):
    """newresampler::smooth_data (R/resampler.cpp:168-230); returns the smoothed D x V rows (and the smoothed mask)."""
    d, pd = _d(np.atleast_2d(data))
    assert d.shape[1] == orig_mesh.V
    out = np.zeros((d.shape[0], sph_low.V))
    if excl is None:
        check(lib().msm_smooth_data(orig_mesh.h, pd, d.shape[0], sph_low.h, float(sigma), None, out.ctypes.data_as(c_dp), None))
        return out
    e, pe = _d(excl)
    eo = np.zeros(sph_low.V)
    check(lib().msm_smooth_data(orig_mesh.h, pd, d.shape[0], sph_low.h, float(sigma), pe, out.ctypes.data_as(c_dp), eo.ctypes.data_as(c_dp)))
    return out, eo


def variance_normalise(data, excl=None):
    """variance_normalise (M/reg_tools.cpp:804-843) of a D x V matrix; returns the normalised copy."""
    out = np.array(np.atleast_2d(data), dtype=np.float64, order="C")
    pe = _d(excl)[1] if excl is not None else None
    check(lib().msm_variance_normalise(out.ctypes.data_as(c_dp), out.shape[0], out.shape[1], pe))
    return out


def mcmc_optimise(unary, tcosts, triplets, labeling, mcparam=0.8, iters=100, seed=0):
    """MCMC::optimise (M/mcmc_opt.h:31-134) over the unary (L x N) and triplet (T x L x L x L) tables; returns the new labeling."""
    U, pu = _d(unary)
    L, N = U.shape
    tc, ptc = _d(tcosts)
    tr = np.ascontiguousarray(triplets, dtype=np.int32)
    T = tr.shape[0]
    assert tc.size == T * L ** 3
    lab = np.array(labeling, dtype=np.int32)
    check(lib().msm_mcmc_optimise(pu, ptc, tr.ctypes.data_as(c_ip), N, L, T, float(mcparam), int(iters), int(seed), lab.ctypes.data_as(c_ip)))
    return lab


def fusion_icm_step(unary2, octets, triplets, passes=5, quads=None, pairs=None):
    """msm_fusion_icm_step: the stand-in binary solve of one label step (iterated conditional modes; NOT ELC + FastPD).  unary2 N x 2
    (current, proposed) or an int N (no unary costs), octets T x 8 / quads P x 4 as tripletOctets / fusionMove return them; returns x (N):
    1 where the proposed label is taken."""
    if np.ndim(unary2) == 0:
        N, u, pu = int(unary2), None, None
    else:
        u = np.ascontiguousarray(unary2, dtype=np.float64)
        N, pu = u.shape[0], u.ctypes.data_as(c_dp)
        assert u.shape == (N, 2)
    tr = np.ascontiguousarray(triplets if triplets is not None else np.zeros((0, 3)), dtype=np.int32)
    e = np.ascontiguousarray(octets if octets is not None else np.zeros((0, 8)), dtype=np.float64)
    pr = np.ascontiguousarray(pairs if pairs is not None else np.zeros((0, 2)), dtype=np.int32)
    q = np.ascontiguousarray(quads if quads is not None else np.zeros((0, 4)), dtype=np.float64)
    T, P = tr.shape[0], pr.shape[0]
    assert e.size == 8 * T and q.size == 4 * P
    x = np.zeros(N, dtype=np.int32)
    check(lib().msm_fusion_icm_step(pu, q.ctypes.data_as(c_dp), pr.ctypes.data_as(c_ip), P, e.ctypes.data_as(c_dp), tr.ctypes.data_as(c_ip), T, N, int(passes),
                                    x.ctypes.data_as(c_ip)))
    return x


def pairwise_icm(unary, paircosts, pairs, labeling=None, passes=100):
    """msm_pairwise_icm: the stand-in solve of the multi-label pairwise MRF of --regoption=1 (iterated conditional modes; NOT FastPD).
    unary L x N, paircosts P x L x L flat as computePairwiseCosts returns it, pairs P x 2; returns the labeling (N)."""
    u = np.ascontiguousarray(unary, dtype=np.float64)
    L, N = u.shape
    pr = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    pc = np.ascontiguousarray(paircosts, dtype=np.float64)
    assert pc.size == pr.shape[0] * L * L
    lab = np.zeros(N, dtype=np.int32) if labeling is None else np.array(labeling, dtype=np.int32)
    check(lib().msm_pairwise_icm(u.ctypes.data_as(c_dp), pc.ctypes.data_as(c_dp), pr.ctypes.data_as(c_ip), N, L, pr.shape[0], int(passes), lab.ctypes.data_as(c_ip)))
    return lab


def nearest_neighbour_interpolation(orig_mesh, data, q, excl=None):
    d, pd = _d(np.atleast_2d(data))
    x, px = _soa(q)
    out = np.zeros((d.shape[0], x.shape[1]))
    if excl is None:
        check(lib().msm_nearest_neighbour(orig_mesh.h, pd, d.shape[0], px, x.shape[1], None, out.ctypes.data_as(c_dp), None))
        return out
    eo = np.zeros(x.shape[1])
    check(lib().msm_nearest_neighbour(orig_mesh.h, pd, d.shape[0], px, x.shape[1], _d(excl)[1], out.ctypes.data_as(c_dp), eo.ctypes.data_as(c_dp)))
    return out, eo


# ------------------------------------------------------------------ cost function
class DiscreteCostFunction:
    """NonLinearSRegDiscreteCostFunction family behind the DiscreteCostFunction evaluator interface."""

    def __init__(self, ctx, kind="univariate", simmeasure=2, rmode=3, lambda_=0.1, mu=0.1, kappa=10.0, k_exp=2.0, rexp=2.0, range_=1.0,
                 percentile=0.75):
        self.ctx = ctx
        self.params = CostParams(KINDS[kind], simmeasure, rmode, 0, lambda_, mu, kappa, k_exp, rexp, range_, percentile)
        self.h = lib().msm_cost_create(ctx.h, C.byref(self.params))
        if not self.h:
            raise MsmError(-1, lib().msm_last_error().decode())
        self._keep = {}

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            lib().msm_cost_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()

    def set_meshes(self, target, source, cpgrid):
        self._keep.update(target=target, source=source, cpgrid=cpgrid)
        self.N = cpgrid.V
        check(lib().msm_cost_set_meshes(self.h, target.h, source.h, cpgrid.h))

    def set_anatomical(self, sphere, atarget_xyz, asource_xyz, asource_tri, w_ptr, w_cp, w_val, face_ptr, face_idx):
        """set_anatomical + set_anatomical_neighbourhood (M/DiscreteCostFunction.h:160-170): sphere = _TARGEThi (a Mesh),
        atarget_xyz = _aTARGET coordinates on the sphere's vertices, asource_* = _aSOURCE, w_* = _ANATbaryweights (CSR over
        _aSOURCE vertices, control point ids ascending), face_* = NEARESTFACES (CSR over triplets)."""
        at, pat = _soa(atarget_xyz)
        asx, pas = _soa(asource_xyz)
        tri, ptri = _tri_soa(asource_tri)
        wp, pwp = _i(w_ptr)
        wc, pwc = _i(w_cp)
        wv, pwv = _d(w_val)
        fp, pfp = _i(face_ptr)
        fi, pfi = _i(face_idx)
        assert at.shape[1] == sphere.V
        self._keep.update(asphere=sphere)
        check(lib().msm_cost_set_anatomical(self.h, sphere.h, pat, pas, asx.shape[1], ptri, tri.shape[1], pwp, pwc, pwv, pfp, pfi))

    def reset_source(self, source):
        self._keep["source"] = source
        check(lib().msm_cost_reset_source(self.h, source.h))

    def reset_CPgrid(self, cpgrid):
        self._keep["cpgrid"] = cpgrid
        check(lib().msm_cost_reset_cpgrid(self.h, cpgrid.h))

    def set_featurespace(self, src_feat, ref_feat=None):
        f, pf = _d(np.atleast_2d(src_feat))
        if ref_feat is not None:
            self._keep["target"].set_pvalues(ref_feat)
        check(lib().msm_cost_set_source_features(self.h, pf, f.shape[0]))

    def set_dataaffintyweighting(self, w):
        if w is None:
            check(lib().msm_cost_set_cfweight(self.h, None, 0))
        else:
            w, pw = _d(np.atleast_2d(w))
            check(lib().msm_cost_set_cfweight(self.h, pw, w.shape[0]))

    def set_spacings(self, maxsep, mvdmax):
        check(lib().msm_cost_set_spacings(self.h, _d(maxsep)[1], float(mvdmax)))

    def set_labels(self, labels, rot):
        l, pl = _soa(labels)
        self.L = l.shape[1]
        check(lib().msm_cost_set_labels(self.h, pl, self.L, _d(rot)[1]))

    def setTriplets(self, triplets):
        t, pt = _i(triplets)
        self.T = len(t)
        check(lib().msm_cost_set_triplets(self.h, pt, self.T))

    def setPairs(self, pairs):
        p, pp = _i(pairs)
        self.P = len(p)
        check(lib().msm_cost_set_pairs(self.h, pp, self.P))

    def get_source_data(self):
        check(lib().msm_cost_get_source_data(self.h))

    def patches(self):
        ng = C.c_int32()
        check(lib().msm_cost_patches(self.h, C.byref(ng), None, None, 0))
        ptr = np.zeros(ng.value + 1, dtype=np.int32)
        check(lib().msm_cost_patches(self.h, C.byref(ng), ptr.ctypes.data_as(c_ip), None, 0))
        idx = np.zeros(max(int(ptr[-1]), 1), dtype=np.int32)
        check(lib().msm_cost_patches(self.h, C.byref(ng), ptr.ctypes.data_as(c_ip), idx.ctypes.data_as(c_ip), len(idx)))
        return ptr, idx[: int(ptr[-1])]

    def absolute_weights(self):
        out = np.zeros(self.N)
        check(lib().msm_cost_absolute_weights(self.h, out.ctypes.data_as(c_dp)))
        return out

    # computeUnaryCosts() + getUnaryCosts(): table[label, node]
    def computeUnaryCosts(self, out=None):
        """computeUnaryCosts() + getUnaryCosts().  `out` (optional, L x N): the array to fill, e.g. Context.host_array((L, N)):
        pinned memory the copy engine writes directly."""
        U = np.zeros((self.L, self.N)) if out is None else out
        assert U.shape == (self.L, self.N) and U.dtype == np.float64 and U.flags.c_contiguous
        check(lib().msm_cost_unary_table(self.h, U.ctypes.data_as(c_dp)))
        return U

    def computeUnaryCosts_async(self):
        check(lib().msm_cost_unary_table_async(self.h))

    def getUnaryCosts(self):
        U = np.zeros((self.L, self.N))
        check(lib().msm_cost_unary_table_fetch(self.h, U.ctypes.data_as(c_dp)))
        return U

    def computeUnaryCost(self, nodes, labels):
        n, pn = _i(np.atleast_1d(nodes))
        l, pl = _i(np.atleast_1d(labels))
        out = np.zeros(len(n))
        check(lib().msm_cost_unary_batch(self.h, pn, pl, len(n), out.ctypes.data_as(c_dp)))
        return out

    def computeTripletCost(self, triplet, la, lb, lc):
        t, pt = _i(np.atleast_1d(triplet))
        a, pa = _i(np.atleast_1d(la))
        b, pb = _i(np.atleast_1d(lb))
        c_, pc = _i(np.atleast_1d(lc))
        out = np.zeros(len(t))
        check(lib().msm_cost_triplet_batch(self.h, pt, pa, pb, pc, len(t), out.ctypes.data_as(c_dp)))
        return out

    def tripletOctets(self, labeling, label, out=None):
        """One label step of Fusion (Fusion.h:181-196).  `out` (optional, T x 8): the array to fill, e.g. Context.host_array((T, 8))
        -- mapped pinned memory the kernels write directly, as the optimiser's per-step buffer would be."""
        lab = labeling if (type(labeling) is np.ndarray and labeling.dtype == np.int32 and labeling.flags.c_contiguous) else np.ascontiguousarray(labeling, dtype=np.int32)
        assert lab.shape == (self.N,)
        if out is None:
            out = np.zeros((self.T, 8))
        assert out.shape == (self.T, 8) and out.dtype == np.float64 and out.flags.c_contiguous
        st = lib().msm_cost_triplet_octets(self.h, lab.ctypes.data, int(label), out.ctypes.data)
        if st:
            check(st)
        return out

    def prefetchTripletOctets(self, labeling, label, out):
        """msm_cost_triplet_octets_prefetch: queue the label step (labeling, label) into `out` (a Context.host_array, T x 8) without waiting -- a hint;
        the next tripletOctets with the same labeling, label and out only waits for it, any other call on this cost function drops it"""
        lab = labeling if (type(labeling) is np.ndarray and labeling.dtype == np.int32 and labeling.flags.c_contiguous) else np.ascontiguousarray(labeling, dtype=np.int32)
        st = lib().msm_cost_triplet_octets_prefetch(self.h, lab.ctypes.data, int(label), out.ctypes.data)
        if st:
            check(st)

    def prefetch_stats(self):
        """(label steps taken from a prefetch, prefetches dropped) since creation"""
        a, b = C.c_int64(), C.c_int64()
        check(lib().msm_cost_prefetch_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def computeTripletCosts(self, t0=0, t1=None, pinned=False):
        """tcosts[t][a][b][c] (M/DiscreteCostFunction.cpp:245-253) for the triplets t0 <= t < t1.  Like the reference's
        tcosts member the table lives in the object: the returned array is reused by the next call of the same shape (a fresh
        281 MB array per call costs more in page faults than the table takes to compute and copy).  pinned=True: the table is delivered into
        a pinned buffer of the CONTEXT instead -- valid until the next pinned table of any cost function on this context."""
        t1 = self.T if t1 is None else t1
        shape = (t1 - t0, self.L, self.L, self.L)
        if pinned:  # one grow-only pinned buffer per context: the 281 MB of an ico4 table arrive at PCIe speed (5 ms; 25 ms into pageable memory)
            out = self.ctx.scratch_host_array("tcosts", shape)
        else:
            out = self._keep.get("tcosts")
            if out is None or out.shape != shape:
                out = self._keep["tcosts"] = np.empty(shape)
        check(lib().msm_cost_triplet_table(self.h, int(t0), int(t1), out.ctypes.data_as(c_dp)))
        return out

    def computePairwiseCost(self, pair, la, lb):
        p, pp = _i(np.atleast_1d(pair))
        a, pa = _i(np.atleast_1d(la))
        b, pb = _i(np.atleast_1d(lb))
        out = np.zeros(len(p))
        check(lib().msm_cost_pairwise_batch(self.h, pp, pa, pb, len(p), out.ctypes.data_as(c_dp)))
        return out

    def computePairwiseCosts(self):
        out = np.zeros(self.P * self.L * self.L)
        check(lib().msm_cost_pairwise_table(self.h, out.ctypes.data_as(c_dp)))
        return out

    def evaluateTotalCostSum(self, labeling):
        lab, pl = _i(labeling)
        tot = C.c_double()
        parts = np.zeros(3)
        check(lib().msm_cost_total(self.h, pl, C.byref(tot), parts.ctypes.data_as(c_dp)))
        return tot.value, parts

    def enable_timing(self, on=True):
        check(lib().msm_cost_enable_timing(self.h, int(on)))

    def kernel_times(self):
        ms = np.zeros(64)
        n = C.c_int32()
        check(lib().msm_cost_kernel_times(self.h, ms.ctypes.data_as(c_dp), 64, C.byref(n)))
        return ms[: n.value].copy()

    def counters(self):
        c = (C.c_int64 * 4)()
        check(lib().msm_cost_counters(self.h, c))
        return dict(samples=c[0], unary=c[1], triplet=c[2], pairwise=c[3])


# ------------------------------------------------------------------ groupwise (gMSM)
class DiscreteGroupCostFunction:
    """DiscreteGroupModel::setupCostFunction + DiscreteGroupCostFunction's evaluators
    (/root/reference/libraries/msm-newmeshreg/src/DiscreteGroupModel.cpp:163-196, DiscreteGroupCostFunction.cpp:26-98)."""

    def __init__(self, ctx, num_subjects, simmeasure=2, fixnan=False, lambda_=0.1, mu=0.1, kappa=10.0, k_exp=2.0, rexp=2.0, range_=1.0,
                 percentile=0.75):
        from ._lib import GroupParams

        self.ctx = ctx
        self.S = num_subjects
        self.params = GroupParams(simmeasure, int(fixnan), lambda_, mu, kappa, k_exp, rexp, range_, percentile)
        self.h = lib().msm_group_create(ctx.h, C.byref(self.params), num_subjects)
        if not self.h:
            raise MsmError(-1, lib().msm_last_error().decode())
        self._keep = {}

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            lib().msm_group_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()

    def set_template(self, mesh, mask=None):
        self._keep["template"] = mesh
        check(lib().msm_group_set_template(self.h, mesh.h, _d(mask)[1] if mask is not None else None))

    def Initialize(self, cp_xyz, cp_tri):
        x, px = _soa(cp_xyz)
        t, pt = _tri_soa(cp_tri)
        self.N, self.Tc = x.shape[1], t.shape[1]
        check(lib().msm_group_set_controlgrid(self.h, px, pt, self.N, self.Tc))

    def reset_meshspace(self, subject, data_mesh, feat):
        f, pf = _d(np.atleast_2d(feat))
        self._keep[("data", subject)] = data_mesh
        self.D = f.shape[0]
        check(lib().msm_group_set_subject(self.h, subject, data_mesh.h, pf, f.shape[0]))

    def reset_CPgrid(self, subject, cp_xyz):
        check(lib().msm_group_reset_cpgrid(self.h, subject, _soa(cp_xyz)[1]))

    def set_labels(self, labels):
        l, pl = _soa(labels)
        self.L = l.shape[1]
        check(lib().msm_group_set_labels(self.h, pl, self.L))

    REFERENCE_ORDER, CP_MAJOR = 0, 1
    DEVICE_ROTATIONS, HOST_ROTATIONS = 0, 1

    def set_rotation_mode(self, mode):
        """who computes the rotation matrices of the data meshes' vertices in get_patch_data: HOST_ROTATIONS (default) -- the host's libm, as the reference
        calls it: the rotated meshes are then the reference's to the bit, which decides the resampled values where a label carries data vertices exactly onto
        template vertices (regular icospheres on both sides) -- or DEVICE_ROTATIONS (msm_group_set_rotation_mode).  Takes effect at the next set-up."""
        check(lib().msm_group_set_rotation_mode(self.h, int(mode)))

    def set_pair_layout(self, layout):
        """the order of the pair list (getPairs and every pair index / range): REFERENCE_ORDER = estimate_pairs' own (subject A, control point, subject
        B), CP_MAJOR = control point by control point along a space-filling curve -- a contiguous slice is then a region of the sphere (the layout
        dist.sharded_group_setup chooses for more than one rank).  Takes effect at the next set-up (msm_group_set_pair_layout)."""
        check(lib().msm_group_set_pair_layout(self.h, int(layout)))

    def setupCostFunction(self):
        check(lib().msm_group_setup(self.h))
        n, p, t = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().msm_group_sizes(self.h, C.byref(n), C.byref(p), C.byref(t)))
        self.num_nodes, self.P, self.T = n.value, p.value, t.value

    # ---- sharded set-up (one process per GPU; see newmsm_amd.dist.sharded_group_setup) ----
    def setup_subjects(self, subjects):
        s, ps = _i(np.asarray(list(subjects), dtype=np.int32))
        check(lib().msm_group_setup_subjects(self.h, ps, len(s)))

    def setup_more_subjects(self, subjects):
        """further subjects of this rank after setup_subjects (msm_group_setup_more_subjects): the set-up in chunks"""
        s, ps = _i(np.asarray(list(subjects), dtype=np.int32))
        check(lib().msm_group_setup_more_subjects(self.h, ps, len(s)))

    def export_subject(self, subject):
        n = C.c_int64()
        check(lib().msm_group_export_subject(self.h, int(subject), None, None, None, 0, C.byref(n)))
        Vt = self._keep["template"].V
        F = np.zeros((self.L, self.D, Vt))
        pptr = np.zeros(self.N * self.L + 1, dtype=np.int32)
        pidx = np.zeros(max(n.value, 1), dtype=np.int32)
        check(lib().msm_group_export_subject(self.h, int(subject), F.ctypes.data_as(c_dp), pptr.ctypes.data_as(c_ip), pidx.ctypes.data_as(c_ip),
                                             len(pidx), C.byref(n)))
        return F, pptr, pidx[: n.value]

    def import_subject(self, subject, F, pptr, pidx):
        F, pF = _d(F)
        pp, ppp = _i(pptr)
        pi, ppi = _i(pidx)
        check(lib().msm_group_import_subject(self.h, int(subject), pF, ppp, ppi if len(pi) else None, len(pi)))

    # device-resident variants: the buffers are torch tensors on this context's GPU (data_ptr()), see newmsm_amd.dist
    def subject_index_count(self, subject):
        n = C.c_int64()
        check(lib().msm_group_export_subject_dev(self.h, int(subject), None, None, None, 0, C.byref(n)))
        return n.value

    def export_subject_dev(self, subject, F_ptr, pptr_ptr, pidx_ptr, cap):
        n = C.c_int64()
        check(lib().msm_group_export_subject_dev(self.h, int(subject), C.c_void_p(F_ptr), C.c_void_p(pptr_ptr), C.c_void_p(pidx_ptr), int(cap), C.byref(n)))
        return n.value

    def import_subject_dev(self, subject, F_ptr, pptr_ptr, pidx_ptr, npidx):
        check(lib().msm_group_import_subject_dev(self.h, int(subject), C.c_void_p(F_ptr), C.c_void_p(pptr_ptr), C.c_void_p(pidx_ptr), int(npidx)))

    def export_subjects_dev(self, subjects, F_ptr, F_stride, pptr_ptr, pptr_stride, pidx_ptr, pidx_stride):
        """subjects[k] into F_ptr + k * F_stride doubles, pptr_ptr + k * pptr_stride, pidx_ptr + k * pidx_stride int32 (device addresses): the send
        buffers of one all-gather, one synchronisation (msm_group_export_subjects_dev); returns the index counts"""
        s, ps = _i(np.asarray(list(subjects), dtype=np.int32))
        n = np.zeros(max(len(s), 1), dtype=np.int64)
        check(lib().msm_group_export_subjects_dev(self.h, ps, len(s), C.c_void_p(F_ptr), int(F_stride), C.c_void_p(pptr_ptr), int(pptr_stride),
                                                  C.c_void_p(pidx_ptr), int(pidx_stride), n.ctypes.data_as(C.POINTER(C.c_int64))))
        return n[: len(s)]

    def import_subjects_dev(self, subjects, F_ptr, F_stride, pptr_ptr, pptr_stride, pidx_ptr, pidx_stride, npidx):
        """the counterpart: subjects[k] out of the receive buffers of one all-gather, range-checked on the device (msm_group_import_subjects_dev)"""
        s, ps = _i(np.asarray(list(subjects), dtype=np.int32))
        n = np.ascontiguousarray(np.asarray(npidx, dtype=np.int64))
        assert len(n) == len(s)
        check(lib().msm_group_import_subjects_dev(self.h, ps, len(s), C.c_void_p(F_ptr), int(F_stride), C.c_void_p(pptr_ptr), int(pptr_stride),
                                                  C.c_void_p(pidx_ptr), int(pidx_stride), n.ctypes.data_as(C.POINTER(C.c_int64))))

    def fusionMove_dev(self, labeling, label, pair_range, triplet_range, quads_ptr, octets_ptr):
        """a slice of a label step, results left in device memory (quads_ptr / octets_ptr: device addresses)"""
        lab, pl = _i(labeling)
        check(lib().msm_group_fusion_move_dev(self.h, pl, int(label), int(pair_range[0]), int(pair_range[1]), int(triplet_range[0]), int(triplet_range[1]),
                                              C.c_void_p(quads_ptr), C.c_void_p(octets_ptr)))

    def time_moves(self, on=True):
        """HIP events around the kernels of every label step (msm_group_time_moves)"""
        check(lib().msm_group_time_moves(self.h, int(on)))

    def move_kernels_ms(self):
        ms = C.c_double(-1.0)
        check(lib().msm_group_move_kernels_ms(self.h, C.byref(ms)))
        return ms.value

    def finalize(self):
        check(lib().msm_group_finalize(self.h))
        n, p, t = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().msm_group_sizes(self.h, C.byref(n), C.byref(p), C.byref(t)))
        self.num_nodes, self.P, self.T = n.value, p.value, t.value

    def getPairs(self):
        out = np.zeros((self.P, 2), dtype=np.int32)
        check(lib().msm_group_get_pairs(self.h, out.ctypes.data_as(c_ip)))
        return out

    def fusionMove(self, labeling, label, out=None):
        """one label step of Fusion::optimize (Fusion.h:157-196): (pair_data buffers P x 4, triplet_data buffers T x 8).
        `out` (optional): the pair (quads, octets) of arrays to fill, e.g. Context.host_array((P, 4)) and ((T, 8)) -- pinned
        memory a copy kernel writes directly."""
        lab, pl = _i(labeling)
        quads, octets = (np.zeros((self.P, 4)), np.zeros((self.T, 8))) if out is None else out
        assert quads.shape == (self.P, 4) and octets.shape == (self.T, 8) and quads.dtype == octets.dtype == np.float64
        assert quads.flags.c_contiguous and octets.flags.c_contiguous
        check(lib().msm_group_fusion_move(self.h, pl, int(label), quads.ctypes.data_as(c_dp), octets.ctypes.data_as(c_dp)))
        return quads, octets

    def getTriplets(self):
        out = np.zeros((self.T, 3), dtype=np.int32)
        check(lib().msm_group_get_triplets(self.h, out.ctypes.data_as(c_ip)))
        return out

    def patch(self, subject, cp, label):
        subject, cp, label = int(subject), int(cp), int(label)
        n = C.c_int32()
        check(lib().msm_group_patch(self.h, subject, cp, label, None, None, 0, C.byref(n)))
        ids = np.zeros(n.value, dtype=np.int32)
        data = np.zeros((n.value, self.D))
        check(lib().msm_group_patch(self.h, subject, cp, label, ids.ctypes.data_as(c_ip), data.ctypes.data_as(c_dp), n.value, C.byref(n)))
        return ids, data

    def computePairwiseCost(self, pair, la, lb):
        p, pp = _i(np.atleast_1d(pair))
        a, pa = _i(np.atleast_1d(la))
        b, pb = _i(np.atleast_1d(lb))
        out = np.zeros(len(p))
        check(lib().msm_group_pairwise_batch(self.h, pp, pa, pb, len(p), out.ctypes.data_as(c_dp)))
        return out

    def computeTripletCost(self, t, la, lb, lc):
        t_, pt = _i(np.atleast_1d(t))
        a, pa = _i(np.atleast_1d(la))
        b, pb = _i(np.atleast_1d(lb))
        c_, pc = _i(np.atleast_1d(lc))
        out = np.zeros(len(t_))
        check(lib().msm_group_triplet_batch(self.h, pt, pa, pb, pc, len(t_), out.ctypes.data_as(c_dp)))
        return out
