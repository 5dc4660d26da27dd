"""ctypes binding of the CPU oracle (oracle/libmsm_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the product package (newmsm_amd).  Parity status: "parity unpinned"
(see oracle/msm_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_lp = C.POINTER(C.c_long)


def build(force=False):
    so = os.path.join(_HERE, "libmsm_oracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-s", "-C", _HERE], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_mesh_create.restype = C.c_void_p
        L.orc_octree_build.restype = C.c_void_p
        L.orc_cost_create.restype = C.c_void_p
        for f in ("orc_compute_area", "orc_dist_to_point", "orc_barycentric_interpolation", "orc_mesh_vertex_area",
                  "orc_mesh_max_vd", "orc_mesh_mean_vd", "orc_corr_weighted", "orc_ssd_weighted", "orc_dice",
                  "orc_gendice", "orc_sim_for_min", "orc_triangle_strain", "orc_triangular_strain", "orc_cost_unary",
                  "orc_cost_triplet", "orc_cost_pairwise", "orc_cost_total"):
            getattr(L, f).restype = C.c_double
        L.orc_adaptive_barycentric_weights.restype = C.c_long
        L.orc_cost_samples.restype = C.c_long
        L.orc_mesh_coords.restype = c_dp
        L.orc_cost_absolute_weights.restype = c_dp
    return _LIB


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(c_ip)


# ---------------------------------------------------------------- geometry
def rotation_matrix(ci, index):
    a, pa = _d(ci)
    b, pb = _d(index)
    R = np.zeros(9)
    st = lib().orc_rotation_matrix(pa, pb, R.ctypes.data_as(c_dp))
    if st:
        raise ValueError("rotation angle is greater than 90 degrees")
    return R.reshape(3, 3)


def project_point(vb, v1, v2, v3):
    out = np.zeros(3)
    lib().orc_project_point(_d(vb)[1], _d(v1)[1], _d(v2)[1], _d(v3)[1], out.ctypes.data_as(c_dp))
    return out


def point_in_triangle(p, a, b, c):
    return bool(lib().orc_point_in_triangle(_d(p)[1], _d(a)[1], _d(b)[1], _d(c)[1]))


def dist_to_point(x0, x1, x2, x3):
    return lib().orc_dist_to_point(_d(x0)[1], _d(x1)[1], _d(x2)[1], _d(x3)[1])


def triangular_strain(orig, final, mu, kappa, k_exp):
    o, po = _d(orig)
    f, pf = _d(final)
    return lib().orc_triangular_strain(po, pf, C.c_double(mu), C.c_double(kappa), C.c_double(k_exp))


def sim_for_min(sim, A, B, w, percentile=0.75):
    A, pa = _d(A)
    B, pb = _d(B)
    w, pw = _d(w)
    return lib().orc_sim_for_min(int(sim), pa, pb, pw, len(A), C.c_double(percentile))


# ---------------------------------------------------------------- icosphere / mesh
def icosphere_counts(order):
    v, t = C.c_int(), C.c_int()
    lib().orc_icosphere_counts(order, C.byref(v), C.byref(t))
    return v.value, t.value


def icosphere(order, radius=100.0, literal=False):
    """make_mesh_from_icosa(order) followed by true_rescale(radius). Returns (xyz[V,3], tri[T,3])."""
    V, T = icosphere_counts(order)
    xyz = np.zeros((V, 3))
    tri = np.zeros((T, 3), dtype=np.int32)
    lib().orc_icosphere(order, int(literal), xyz.ctypes.data_as(c_dp), tri.ctypes.data_as(c_ip))
    if radius is not None:
        lib().orc_true_rescale(xyz.ctypes.data_as(c_dp), V, C.c_double(radius))
    return xyz, tri


def resample_anatomy_grid(cp_xyz, cp_tri, levels, rad=100.0, literal=False):
    """Mesh_registration::resample_anatomy (M/mesh_registration.cpp:250-332) without its surface_resample call: dict(sphere_xyz, sphere_tri, w_ptr, w_cp,
    w_val, face_ptr, face_idx) -- ANAT_ico, _ANATbaryweights and NEARESTFACES (CSR) as the reference builds them."""
    x, px = _d(cp_xyz)
    t, pt = _i(cp_tri)
    va, ta = C.c_int(), C.c_int()
    lib().orc_resample_anatomy_sizes(len(x), len(t), int(levels), C.byref(va), C.byref(ta))
    Va, Ta = va.value, ta.value
    axyz, atri = np.zeros((Va, 3)), np.zeros((Ta, 3), dtype=np.int32)
    w_ptr, w_cp, w_val = np.zeros(Va + 1, dtype=np.int32), np.zeros(3 * Va, dtype=np.int32), np.zeros(3 * Va)
    face_ptr, face_idx = np.zeros(len(t) + 1, dtype=np.int32), np.zeros(Ta, dtype=np.int32)
    st = lib().orc_resample_anatomy_grid(px, len(x), pt, len(t), int(levels), C.c_double(rad), int(literal), axyz.ctypes.data_as(c_dp), atri.ctypes.data_as(c_ip),
                                         w_ptr.ctypes.data_as(c_ip), w_cp.ctypes.data_as(c_ip), w_val.ctypes.data_as(c_dp), face_ptr.ctypes.data_as(c_ip),
                                         face_idx.ctypes.data_as(c_ip))
    if st:
        raise RuntimeError("orc_resample_anatomy_grid failed")
    n = int(w_ptr[-1])
    return dict(sphere_xyz=axyz, sphere_tri=atri, w_ptr=w_ptr, w_cp=w_cp[:n].copy(), w_val=w_val[:n].copy(), face_ptr=face_ptr, face_idx=face_idx)


def surface_resample(anat_xyz, sph_tree, q_xyz):
    """newresampler::surface_resample / project_anatomical_mesh (R/resampler.cpp:284-302, :260-282): get_barycentric_weights of the points q in the sphere
    the anatomy lives on (its octree: sph_tree), then newPt += anat(id) * w in std::map order (ascending vertex id) from Point() = 0."""
    st, _, vid, w = sph_tree.barycentric_weights(q_xyz)
    if np.any(st != 0):
        raise RuntimeError("octree query failed")
    anat = np.asarray(anat_xyz, dtype=np.float64)
    key = np.argsort(vid, axis=1, kind="stable")
    vid, w = np.take_along_axis(vid, key, axis=1), np.take_along_axis(w, key, axis=1)
    out = np.zeros((len(vid), 3))
    for k in range(3):
        out = out + anat[vid[:, k]] * w[:, k][:, None]
    return out


class Mesh:
    def __init__(self, xyz, tri):
        self.xyz, px = _d(xyz)
        self.tri, pt = _i(tri)
        self.V, self.T = len(self.xyz), len(self.tri)
        self.h = C.c_void_p(lib().orc_mesh_create(px, self.V, pt, self.T))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_mesh_destroy(self.h)
            self.h = None

    def set_coords(self, xyz, refresh_areas=True):
        self.xyz, px = _d(xyz)
        lib().orc_mesh_set_coords(self.h, px, int(refresh_areas))

    def adjacency(self):
        np_, n_, tp_, t_ = c_ip(), c_ip(), c_ip(), c_ip()
        lib().orc_mesh_adjacency(self.h, C.byref(np_), C.byref(n_), C.byref(tp_), C.byref(t_))
        nbr_ptr = np.ctypeslib.as_array(np_, (self.V + 1,)).copy()
        tid_ptr = np.ctypeslib.as_array(tp_, (self.V + 1,)).copy()
        nbr = np.ctypeslib.as_array(n_, (int(nbr_ptr[-1]),)).copy()
        tid = np.ctypeslib.as_array(t_, (int(tid_ptr[-1]),)).copy()
        return nbr_ptr, nbr, tid_ptr, tid

    def vertex_areas(self):
        return np.array([lib().orc_mesh_vertex_area(self.h, v) for v in range(self.V)])

    def max_vd(self):
        return lib().orc_mesh_max_vd(self.h)


class Octree:
    def __init__(self, mesh):
        self.mesh = mesh
        self.h = C.c_void_p(lib().orc_octree_build(mesh.h))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_octree_destroy(self.h)
            self.h = None

    def stats(self):
        s = (C.c_long * 5)()
        lib().orc_octree_stats(self.h, s)
        return dict(nodes=s[0], leaves=s[1], depth=s[2], refs=s[3], max_leaf=s[4])

    def closest_triangle(self, pts, count_tests=False):
        pts, _ = _d(pts)
        pts = pts.reshape(-1, 3)
        out = np.zeros(len(pts), dtype=np.int32)
        n = C.c_long(0)
        f = lib().orc_octree_closest_triangle
        for k in range(len(pts)):
            out[k] = f(self.h, pts[k].ctypes.data_as(c_dp), C.byref(n) if count_tests else None)
        return (out, n.value) if count_tests else out

    def closest_vertex(self, pts):
        pts, _ = _d(pts)
        pts = pts.reshape(-1, 3)
        f = lib().orc_octree_closest_vertex
        return np.array([f(self.h, pts[k].ctypes.data_as(c_dp)) for k in range(len(pts))], dtype=np.int32)

    def barycentric_weights(self, q, raw=False):
        q, pq = _d(q)
        N = len(q)
        tri = np.zeros(N, dtype=np.int32)
        vid = np.zeros((N, 3), dtype=np.int32)
        w = np.zeros((N, 3))
        f = lib().orc_barycentric_weights_raw if raw else lib().orc_barycentric_weights
        st = f(self.h, pq, N, tri.ctypes.data_as(c_ip), vid.ctypes.data_as(c_ip), w.ctypes.data_as(c_dp))
        return st, tri, vid, w


# ---------------------------------------------------------------- resampler
def adaptive_barycentric_weights(in_mesh, new_mesh, excl=None):
    pe = _d(excl)[1] if excl is not None else None
    nnz = lib().orc_adaptive_barycentric_weights(in_mesh.h, new_mesh.h, pe, None, None, None)
    if nnz < 0:
        raise RuntimeError("octree query failed")
    rp = np.zeros(new_mesh.V + 1, dtype=np.int32)
    col = np.zeros(nnz, dtype=np.int32)
    val = np.zeros(nnz)
    lib().orc_adaptive_barycentric_weights(in_mesh.h, new_mesh.h, pe, rp.ctypes.data_as(c_ip), col.ctypes.data_as(c_ip),
                                           val.ctypes.data_as(c_dp))
    return rp, col, val


def apply_weights(rp, col, val, data, excl=None):
    data, pd = _d(np.atleast_2d(data))
    D, Vin = data.shape
    N = len(rp) - 1
    out = np.zeros((D, N))
    pe = _d(excl)[1] if excl is not None else None
    lib().orc_apply_weights(_i(rp)[1], _i(col)[1], _d(val)[1], N, pd, D, Vin, pe, out.ctypes.data_as(c_dp))
    return out


def metric_resample(in_mesh, data, new_mesh):
    data, pd = _d(np.atleast_2d(data))
    out = np.zeros((data.shape[0], new_mesh.V))
    st = lib().orc_metric_resample(in_mesh.h, pd, data.shape[0], new_mesh.h, out.ctypes.data_as(c_dp))
    if st:
        raise RuntimeError("octree query failed")
    return out


def metric_resample_excl(in_mesh, data, new_mesh, excl):
    data, pd = _d(np.atleast_2d(data))
    out, eo = np.zeros((data.shape[0], new_mesh.V)), np.zeros(new_mesh.V)
    st = lib().orc_metric_resample_excl(in_mesh.h, pd, data.shape[0], new_mesh.h, _d(excl)[1], out.ctypes.data_as(c_dp), eo.ctypes.data_as(c_dp))
    if st:
        raise RuntimeError("octree query failed")
    return out, eo


def create_exclusion(data, thrl, thru):
    data, pd = _d(np.atleast_2d(data))
    out = np.zeros(data.shape[1])
    lib().orc_create_exclusion(pd, data.shape[0], data.shape[1], C.c_double(thrl), C.c_double(thru), out.ctypes.data_as(c_dp))
    return out


def nearest_neighbour_excl(orig_mesh, data, q, excl):
    data, pd = _d(np.atleast_2d(data))
    q, pq = _d(q)
    out, eo = np.zeros((data.shape[0], len(q))), np.zeros(len(q))
    st = lib().orc_nearest_neighbour_excl(orig_mesh.h, pd, data.shape[0], pq, len(q), _d(excl)[1], out.ctypes.data_as(c_dp), eo.ctypes.data_as(c_dp))
    if st:
        raise RuntimeError("octree query failed")
    return out, eo


def sphere_project_warp(sphere, from_mesh, to_xyz):
    s = np.array(sphere, dtype=np.float64, order="C")
    to, pt = _d(to_xyz)
    st = lib().orc_sphere_project_warp(s.ctypes.data_as(c_dp), len(s), from_mesh.h, pt)
    if st:
        raise RuntimeError("octree query failed")
    return s


def nearest_neighbour(orig_mesh, data, q):
    data, pd = _d(np.atleast_2d(data))
    q, pq = _d(q)
    out = np.zeros((data.shape[0], len(q)))
    st = lib().orc_nearest_neighbour(orig_mesh.h, pd, data.shape[0], pq, len(q), out.ctypes.data_as(c_dp))
    if st:
        raise RuntimeError("octree query failed")
    return out


def smooth_data(orig_mesh, data, sph_low, sigma, excl=None):
    data, pd = _d(np.atleast_2d(data))
    out = np.zeros((data.shape[0], sph_low.V))
    eo = np.zeros(sph_low.V) if excl is not None else None
    pe = _d(excl)[1] if excl is not None else None
    st = lib().orc_smooth_data(orig_mesh.h, pd, data.shape[0], sph_low.h, C.c_double(sigma), pe, out.ctypes.data_as(c_dp),
                               eo.ctypes.data_as(c_dp) if eo is not None else None)
    if st:
        raise RuntimeError("smooth_data failed (%d)" % st)
    return (out, eo) if excl is not None else out


def unfold(mesh, rad=100.0):
    """unfold (reg_tools.cpp:131-178) on the mesh's coordinates in place; returns (passes, folded vertices of the first pass)."""
    L = lib()
    L.orc_mesh_coords.restype = c_dp
    first = C.c_int()
    passes = L.orc_unfold(mesh.h, C.c_double(rad), C.byref(first))
    if passes < 0:
        raise RuntimeError("get_triangle: index exceeds face dimensions")
    mesh.xyz = np.ctypeslib.as_array(L.orc_mesh_coords(mesh.h), (mesh.V, 3)).copy()
    return passes, first.value


def variance_normalise(data, excl=None):
    """variance_normalise (reg_tools.cpp:804-843) of a D x V matrix; returns the normalised copy."""
    out = np.array(np.atleast_2d(data), dtype=np.float64, order="C")
    pe = _d(excl)[1] if excl is not None else None
    lib().orc_variance_normalise(out.ctypes.data_as(c_dp), out.shape[0], out.shape[1], pe)
    return out


# ---------------------------------------------------------------- discrete model host logic
def cp_spacings(cp_mesh):
    ms = np.zeros(cp_mesh.V)
    mvd = C.c_double()
    lib().orc_cp_spacings(cp_mesh.h, ms.ctypes.data_as(c_dp), C.byref(mvd))
    return ms, mvd.value


def label_sampling_grid(sg_mesh, max_dist, abs_is_int=False, maxn=4096):
    s = np.zeros((maxn, 3))
    b = np.zeros((maxn, 3))
    ns, nb, cen = C.c_int(), C.c_int(), C.c_int()
    st = lib().orc_label_sampling_grid(sg_mesh.h, C.c_double(max_dist), int(abs_is_int), C.byref(cen), s.ctypes.data_as(c_dp),
                                       C.byref(ns), b.ctypes.data_as(c_dp), C.byref(nb), maxn)
    if st:
        raise RuntimeError("too many labels")
    return cen.value, s[: ns.value].copy(), b[: nb.value].copy()


def rescale_sampling_grid(samples, scale):
    s, ps = _d(samples)
    sc = C.c_double(scale)
    out = np.zeros_like(s)
    lib().orc_rescale_sampling_grid(ps, len(s), C.byref(sc), out.ctypes.data_as(c_dp))
    return out, sc.value


def cp_rotations(centre, cp_xyz):
    cp, pc = _d(cp_xyz)
    rot = np.zeros((len(cp), 9))
    lib().orc_cp_rotations(_d(centre)[1], pc, len(cp), rot.ctypes.data_as(c_dp))
    return rot


def estimate_triplets(cp_mesh):
    t = np.zeros((cp_mesh.T, 3), dtype=np.int32)
    lib().orc_estimate_triplets(cp_mesh.h, t.ctypes.data_as(c_ip))
    return t


def estimate_pairs(cp_mesh):
    n = lib().orc_estimate_pairs(cp_mesh.h, None)
    p = np.zeros((n, 2), dtype=np.int32)
    lib().orc_estimate_pairs(cp_mesh.h, p.ctypes.data_as(c_ip))
    return p


# ---------------------------------------------------------------- cost function
class CostParams(C.Structure):
    _fields_ = [("kind", C.c_int), ("simmeasure", C.c_int), ("rmode", C.c_int), ("lambda_", C.c_double),
                ("mu", C.c_double), ("kappa", C.c_double), ("k_exp", C.c_double), ("rexp", C.c_double),
                ("range", C.c_double), ("percentile", C.c_double)]


KINDS = dict(univariate=0, multivariate=1, patchwise=2, ho_univariate=3, ho_multivariate=4)


class Cost:
    """Mirror of NonLinearSRegDiscreteCostFunction and its subclasses (oracle side)."""

    def __init__(self, kind="univariate", simmeasure=2, rmode=3, lambda_=0.1, mu=0.1, kappa=10.0, k_exp=2.0, rexp=2.0,
                 range_=1.0, percentile=0.75):
        self.params = CostParams(KINDS[kind], simmeasure, rmode, lambda_, mu, kappa, k_exp, rexp, range_, percentile)
        self.h = C.c_void_p(lib().orc_cost_create(C.byref(self.params)))
        self._keep = {}

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_cost_destroy(self.h)
            self.h = None

    def set_meshes(self, target, ttree, source, cpgrid):
        self._keep.update(target=target, ttree=ttree, source=source, cpgrid=cpgrid)
        self.N = cpgrid.V
        lib().orc_cost_set_meshes(self.h, target.h, ttree.h, source.h, cpgrid.h)

    def set_anatomical(self, sphere, sphere_tree, atarget_xyz, asource, w_ptr, w_cp, w_val, face_ptr, face_idx):
        at, pat = _d(atarget_xyz)
        wp, pwp = _i(w_ptr)
        wc, pwc = _i(w_cp)
        wv, pwv = _d(w_val)
        fp, pfp = _i(face_ptr)
        fi, pfi = _i(face_idx)
        self._keep.update(asphere=sphere, atree=sphere_tree, at=at, asource=asource, wp=wp, wc=wc, wv=wv, fp=fp, fi=fi)
        lib().orc_cost_set_anatomical(self.h, sphere.h, sphere_tree.h, pat, asource.h, pwp, pwc, pwv, pfp, pfi)

    def reset_source(self, source):
        self._keep["source"] = source
        lib().orc_cost_reset_source(self.h, source.h)

    def reset_cpgrid(self, cpgrid):
        self._keep["cpgrid"] = cpgrid
        lib().orc_cost_reset_cpgrid(self.h, cpgrid.h)

    def set_features(self, src_feat, ref_feat):
        s, ps = _d(np.atleast_2d(src_feat))
        r, pr = _d(np.atleast_2d(ref_feat))
        self._keep.update(sf=s, rf=r)
        self.D = s.shape[0]
        lib().orc_cost_set_features(self.h, ps, pr, self.D)

    def set_cfweight(self, w):
        if w is None:
            lib().orc_cost_set_cfweight(self.h, None, 0)
            return
        w, pw = _d(np.atleast_2d(w))
        self._keep["cfw"] = w
        lib().orc_cost_set_cfweight(self.h, pw, w.shape[0])

    def set_spacings(self, maxsep, mvdmax):
        lib().orc_cost_set_spacings(self.h, _d(maxsep)[1], C.c_double(mvdmax))

    def set_labels(self, labels, rot):
        l, pl = _d(labels)
        r, pr = _d(rot)
        self.L = len(l)
        lib().orc_cost_set_labels(self.h, pl, self.L, pr)

    def set_triplets(self, trip):
        t, pt = _i(trip)
        self.T = len(t)
        lib().orc_cost_set_triplets(self.h, pt, self.T)

    def set_pairs(self, pairs):
        p, pp = _i(pairs)
        self.P = len(p)
        lib().orc_cost_set_pairs(self.h, pp, self.P)

    def get_source_data(self):
        st = lib().orc_cost_get_source_data(self.h)
        if st:
            raise RuntimeError("get_source_data failed (%d)" % st)

    def patches(self):
        ptr, idx, ng = c_ip(), c_ip(), C.c_int()
        lib().orc_cost_patches(self.h, C.byref(ptr), C.byref(idx), C.byref(ng))
        p = np.ctypeslib.as_array(ptr, (ng.value + 1,)).copy()
        i = np.ctypeslib.as_array(idx, (max(int(p[-1]), 1),)).copy()[: int(p[-1])]
        return p, i

    def absolute_weights(self):
        return np.ctypeslib.as_array(lib().orc_cost_absolute_weights(self.h), (self.N,)).copy()

    def unary(self, node, label):
        return lib().orc_cost_unary(self.h, int(node), int(label))

    def unary_table(self, threads=0):
        U = np.zeros((self.L, self.N))
        if threads and threads > 0:
            lib().orc_cost_unary_table_omp(self.h, U.ctypes.data_as(c_dp), int(threads))
        else:
            lib().orc_cost_unary_table(self.h, U.ctypes.data_as(c_dp))
        return U

    def triplet(self, t, la, lb, lc):
        return lib().orc_cost_triplet(self.h, int(t), int(la), int(lb), int(lc))

    def triplet_octets(self, labeling, label, threads=1):
        """One fusion move (I/Fusion/Fusion.h:181-196): E[t, k], k = 000..111 over (A,B,C)."""
        lab, pl = _i(labeling)
        E = np.zeros((self.T, 8))
        lib().orc_cost_triplet_octets(self.h, pl, int(label), E.ctypes.data_as(c_dp), int(threads))
        return E

    def triplet_table(self, t0=0, t1=None):
        t1 = self.T if t1 is None else t1
        out = np.zeros((t1 - t0, self.L, self.L, self.L))
        lib().orc_cost_triplet_table(self.h, int(t0), int(t1), out.ctypes.data_as(c_dp))
        return out

    def pairwise(self, p, la, lb):
        return lib().orc_cost_pairwise(self.h, int(p), int(la), int(lb))

    def pairwise_table(self):
        """paircosts[(pair * L + labelB) * L + labelA], computePairwiseCosts M/DiscreteCostFunction.cpp:228-234"""
        L, P = self.L, self.P
        out = np.zeros(P * L * L)
        lib().orc_cost_pairwise_table(self.h, out.ctypes.data_as(c_dp))
        return out

    def total(self, labeling):
        lab, pl = _i(labeling)
        parts = np.zeros(3)
        tot = lib().orc_cost_total(self.h, pl, parts.ctypes.data_as(c_dp))
        return tot, parts

    def samples(self):
        return lib().orc_cost_samples(self.h)


# ---------------------------------------------------------------- groupwise (gMSM)
class GroupParams(C.Structure):
    _fields_ = [("simmeasure", C.c_int), ("fixnan", C.c_int), ("lambda_", C.c_double), ("mu", C.c_double), ("kappa", C.c_double),
                ("k_exp", C.c_double), ("rexp", C.c_double), ("range", C.c_double), ("percentile", C.c_double)]


class Group:
    """Oracle mirror of DiscreteGroupModel + DiscreteGroupCostFunction."""

    def __init__(self, num_subjects, simmeasure=2, fixnan=False, lambda_=0.1, mu=0.1, kappa=10.0, k_exp=2.0, rexp=2.0, range_=1.0,
                 percentile=0.75):
        L = lib()
        L.orc_group_create.restype = C.c_void_p
        L.orc_group_pairwise.restype = C.c_double
        L.orc_group_triplet.restype = C.c_double
        L.orc_group_pairs.restype = c_ip
        L.orc_group_triplets.restype = c_ip
        self.params = GroupParams(simmeasure, int(fixnan), lambda_, mu, kappa, k_exp, rexp, range_, percentile)
        self.S = num_subjects
        self.h = C.c_void_p(L.orc_group_create(C.byref(self.params), num_subjects))
        self._keep = {}

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_group_destroy(self.h)
            self.h = None

    def set_template(self, mesh, mask=None):
        self._keep["template"] = mesh
        self.Vt = mesh.V
        pm = None
        if mask is not None:
            self._keep["mask"], pm = _d(mask)
        lib().orc_group_set_template(self.h, mesh.h, pm)

    def set_controlgrid(self, cp_mesh):
        self._keep["cp"] = cp_mesh
        self.N, self.Tc = cp_mesh.V, cp_mesh.T
        lib().orc_group_set_controlgrid(self.h, cp_mesh.h)

    def set_subject(self, s, data_mesh, feat):
        f, pf = _d(np.atleast_2d(feat))
        self._keep[("data", s)] = data_mesh
        self.D = f.shape[0]
        lib().orc_group_set_subject(self.h, s, data_mesh.h, pf, f.shape[0])

    def reset_cpgrid(self, s, xyz):
        lib().orc_group_reset_cpgrid(self.h, s, _d(xyz)[1])

    def set_labels(self, labels):
        l, pl = _d(labels)
        self.L = len(l)
        lib().orc_group_set_labels(self.h, pl, self.L)

    def setup(self):
        st = lib().orc_group_setup(self.h)
        if st:
            raise RuntimeError("group setup failed (%d)" % st)
        n, p, t = C.c_int(), C.c_int(), C.c_int()
        lib().orc_group_sizes(self.h, C.byref(n), C.byref(p), C.byref(t))
        self.num_nodes, self.P, self.T = n.value, p.value, t.value

    def set_threads(self, n):
        """OpenMP threads of get_patch_data's loop over the subjects (the reference: num_threads(_nthreads), M/DiscreteGroupModel.cpp:92)"""
        lib().orc_group_set_threads(self.h, int(n))

    def pairwise_batch(self, pair, la, lb, threads=1):
        """computePairwiseCost for n (pair, labelA, labelB) over OpenMP threads, as Fusion::optimize's pair loop runs them (I/Fusion/Fusion.h:164)"""
        p, pp = _i(pair)
        a, pa = _i(la)
        b, pb = _i(lb)
        out = np.empty(len(p))
        lib().orc_group_pairwise_batch(self.h, pp, pa, pb, len(p), out.ctypes.data_as(c_dp), int(threads))
        return out

    def triplet_batch(self, t, la, lb, lc, threads=1):
        tt, pt = _i(t)
        a, pa = _i(la)
        b, pb = _i(lb)
        c, pc = _i(lc)
        out = np.empty(len(tt))
        lib().orc_group_triplet_batch(self.h, pt, pa, pb, pc, len(tt), out.ctypes.data_as(c_dp), int(threads))
        return out

    def pairs(self):
        return np.ctypeslib.as_array(lib().orc_group_pairs(self.h), (self.P, 2)).copy()

    def triplets(self):
        return np.ctypeslib.as_array(lib().orc_group_triplets(self.h), (self.T, 3)).copy()

    def patch(self, s, v, l, cap=4096):
        ids = np.zeros(cap, dtype=np.int32)
        data = np.zeros((cap, self.D))
        n = lib().orc_group_patch(self.h, int(s), int(v), int(l), ids.ctypes.data_as(c_ip), data.ctypes.data_as(c_dp), cap)
        return ids[:n].copy(), data[:n].copy()

    def pairwise(self, pair, la, lb):
        return lib().orc_group_pairwise(self.h, int(pair), int(la), int(lb))

    def triplet(self, t, la, lb, lc):
        return lib().orc_group_triplet(self.h, int(t), int(la), int(lb), int(lc))
