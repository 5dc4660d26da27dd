"""Test-side glue: builds ORACLE objects (oracle/ is test infrastructure) from the numpy inputs that
newmsm_amd.problem prepares for the product path."""
import os

import numpy as np

from oracle import oracle as O


def oracle_cost(inp, kind="univariate", simmeasure=2, rmode=3, **params):
    target = O.Mesh(inp["target_xyz"], inp["target_tri"])
    ttree = O.Octree(target)
    source = O.Mesh(inp["source_orig_xyz"], inp["source_tri"])
    cpgrid = O.Mesh(inp["cp_orig_xyz"], inp["cp_tri"])
    c = O.Cost(kind, simmeasure=simmeasure, rmode=rmode, **params)
    c.set_meshes(target, ttree, source, cpgrid)   # captures _ORIG / _oCPgrid
    source.set_coords(inp["source_xyz"])
    cpgrid.set_coords(inp["cp_xyz"])
    c.reset_source(source)
    c.reset_cpgrid(cpgrid)
    c.set_features(inp["src_feat"], inp["ref_feat"])
    c.set_spacings(inp["maxsep"], inp["mvdmax"])
    c.set_labels(inp["labels"], inp["rot"])
    c.set_triplets(inp["triplets"])
    c.set_pairs(inp["pairs"])
    return c


def ulp_close(a, b, rtol=1e-11, atol=1e-12):
    a = np.asarray(a)
    b = np.asarray(b)
    return np.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)


def oracle_anatomy(cp_order=2, anat_order=4):
    """Anatomical-regulariser inputs (control grid, anatomical sphere, _ANATbaryweights, NEARESTFACES) built with the oracle only, as
    Mesh_registration::resample_anatomy (M/mesh_registration.cpp:250-332) prepares them (O.resample_anatomy_grid)."""
    cxyz, ctri = O.icosphere(cp_order)
    g = O.resample_anatomy_grid(cxyz, ctri, anat_order - cp_order)
    return cxyz, ctri, g["sphere_xyz"], g["sphere_tri"], g["w_ptr"], g["w_cp"], g["w_val"], g["face_ptr"], g["face_idx"]


def oracle_anatomy_by_queries(cp_order=2, anat_order=4):
    """The shape of the same inputs from octree queries instead of the retessellation bookkeeping (NEARESTFACES by face centroid, weights by a search of
    every vertex): what the committed golden vectors of the anatomical strain (tests/golden/oracle_more.npz, g8_*) were generated from in round 1."""
    cxyz, ctri = O.icosphere(cp_order)
    axyz, atri = O.icosphere(anat_order)
    cp = O.Mesh(cxyz, ctri)
    tree = O.Octree(cp)
    cen = axyz[atri].mean(axis=1)
    cen = cen / np.linalg.norm(cen, axis=1, keepdims=True) * 100.0
    ftri = tree.closest_triangle(cen)
    face_ptr = np.zeros(len(ctri) + 1, dtype=np.int32)
    np.add.at(face_ptr, ftri + 1, 1)
    face_ptr = np.cumsum(face_ptr).astype(np.int32)
    face_idx = np.argsort(ftri, kind="stable").astype(np.int32)
    _, _, vid, w = tree.barycentric_weights(axyz)
    key = np.argsort(vid, axis=1, kind="stable")
    w_cp = np.take_along_axis(vid, key, axis=1).astype(np.int32).ravel()
    w_val = np.take_along_axis(w, key, axis=1).ravel()
    w_ptr = (3 * np.arange(len(axyz) + 1)).astype(np.int32)
    return cxyz, ctri, axyz, atri, w_ptr, w_cp, w_val, face_ptr, face_idx


ORACLE_THREADS = int(os.environ.get("MSM_ORACLE_THREADS", "8"))  # OpenMP threads of the oracle's table / octet evaluations


class OracleOps:
    """The calls of newmsm_amd.registration.run_discrete_level answered by the oracle (same methods as ProductOps), so that the
    very same caller loop drives the CPU restatement.  `optimiser` is the caller's optimiser (not part of the path): the
    tests pass the library's host-side Monte Carlo routine to both runs."""

    def __init__(self, optimiser):
        self.optimiser = optimiser

    def icosphere(self, order):
        return O.icosphere(order)

    def mesh(self, xyz, tri, feat=None):
        m = O.Mesh(xyz, tri)
        m.feat = feat
        return m

    def set_coords(self, mesh, xyz):
        mesh.set_coords(xyz)

    def coords(self, mesh):
        return np.array(mesh.xyz)

    def unfold(self, mesh):
        return O.unfold(mesh, 100.0)

    def sphere_project_warp(self, sphere_xyz, from_mesh, to_xyz):
        return O.sphere_project_warp(sphere_xyz, from_mesh, to_xyz)

    def warp_mesh(self, mesh, from_mesh, to_xyz):
        mesh.set_coords(O.sphere_project_warp(np.array(mesh.xyz), from_mesh, to_xyz))

    def metric_resample(self, in_mesh, data, new_mesh, slot=None):
        return O.metric_resample(in_mesh, data, new_mesh)

    def smooth_data(self, mesh, data, sigma):
        return O.smooth_data(mesh, data, mesh, sigma)

    def variance_normalise(self, data):
        return O.variance_normalise(data)

    def nearest_neighbour(self, mesh, data, q_xyz):
        return O.nearest_neighbour(mesh, data, q_xyz)

    def resample_anatomy_grid(self, cp_xyz, cp_tri, levels):
        return O.resample_anatomy_grid(cp_xyz, cp_tri, levels, 100.0)

    def surface_resample(self, anat_xyz, sphere_mesh, q_xyz):
        return O.surface_resample(anat_xyz, O.Octree(sphere_mesh), q_xyz)

    def cp_spacings(self, mesh, xyz, tri):
        return O.cp_spacings(mesh)

    def estimate_triplets(self, mesh, tri):
        return O.estimate_triplets(mesh)

    def estimate_pairs(self, mesh, tri, nodes):
        return O.estimate_pairs(mesh)

    def label_sampling_grid(self, sg_order, max_dist):
        _, s, b = O.label_sampling_grid(O.Mesh(*O.icosphere(sg_order)), max_dist)
        return s, b

    def rescale_sampling_grid(self, samples, scale):
        return O.rescale_sampling_grid(samples, scale)

    def cp_rotations(self, centre, cp_xyz):
        return O.cp_rotations(centre, cp_xyz)

    def cost(self, kind, simmeasure, rmode, params, target, source, cpgrid, src_feat):
        c = O.Cost(kind, simmeasure=simmeasure, rmode=rmode, **params)
        c.set_meshes(target, O.Octree(target), source, cpgrid)
        c.set_features(src_feat, target.feat)
        c.set_pairs(np.zeros((0, 2), dtype=np.int32))
        return _OracleCost(c)

    def mcmc(self, unary, tcosts, triplets, labeling, mcparam, iters, seed):
        return self.optimiser(unary, tcosts, triplets, labeling, mcparam=mcparam, iters=iters, seed=seed)

    def fusion_step(self, unary2, octets, triplets, passes, quads=None, pairs=None):
        from newmsm_amd import api  # the caller's stand-in solver (not part of the path), the same for both runs

        return api.fusion_icm_step(unary2, octets, triplets, passes, quads=quads, pairs=pairs)

    def pairwise_solve(self, unary, paircosts, pairs, passes):
        from newmsm_amd import api  # the caller's stand-in solver (not part of the path), the same for both runs

        return api.pairwise_icm(unary, paircosts, pairs, passes=passes)

    def group(self, S, simmeasure, lambda_, fixnan, **params):
        return _OracleGroup(O.Group(S, simmeasure=simmeasure, lambda_=lambda_, fixnan=fixnan, **params))


class _OracleGroup:
    """newmsm_amd.group_registration's group object answered by the oracle (scalar evaluators: small groups only)"""

    def __init__(self, g):
        self.g, self.keep = g, []

    def set_template(self, mesh, mask=None):
        self.keep.append(mesh)
        self.g.set_template(mesh, mask)

    def initialize(self, cp_mesh, cp_xyz, cp_tri):
        self.keep.append(cp_mesh)
        self.g.set_controlgrid(cp_mesh)

    def set_subject(self, s, mesh, feat):
        self.keep.append(mesh)
        self.g.set_subject(s, mesh, feat)

    def reset_cpgrid(self, s, xyz):
        self.g.reset_cpgrid(s, xyz)

    def set_labels(self, labels):
        self.g.set_labels(labels)

    def setup(self):
        self.g.setup()

    def pairs(self):
        return self.g.pairs()

    def triplets(self):
        return self.g.triplets()

    def fusion_move(self, labeling, label):
        pr, tr = self.g.pairs(), self.g.triplets()
        quads = np.empty((len(pr), 4))
        for p in range(len(pr)):
            a, b = int(labeling[pr[p, 0]]), int(labeling[pr[p, 1]])
            quads[p] = [self.g.pairwise(p, a, b), self.g.pairwise(p, a, label), self.g.pairwise(p, label, b), self.g.pairwise(p, label, label)]
        octets = np.empty((len(tr), 8))
        for t in range(len(tr)):
            cur = [int(labeling[v]) for v in tr[t]]
            for k in range(8):
                octets[t, k] = self.g.triplet(t, *[label if k >> (2 - j) & 1 else cur[j] for j in range(3)])
        return quads, octets


class _OracleCost:
    def __init__(self, c):
        self.c = c

    def reset_source(self, mesh):
        self.c.reset_source(mesh)

    def reset_cpgrid(self, mesh):
        self.c.reset_cpgrid(mesh)

    def set_spacings(self, maxsep, mvdmax):
        self.c.set_spacings(maxsep, mvdmax)

    def set_cfweight(self, w):
        self.c.set_cfweight(w)

    def set_labels(self, labels, rot):
        self.c.set_labels(labels, rot)

    def set_triplets(self, triplets):
        self.c.set_triplets(triplets)
        self.c.set_pairs(np.zeros((0, 2), dtype=np.int32))

    def set_pairs(self, pairs):
        self.c.set_pairs(pairs)
        self.c.set_triplets(np.zeros((0, 3), dtype=np.int32))

    def set_anatomical(self, sphere_mesh, atarget_xyz, asource_xyz, grid):
        self.keep = (sphere_mesh, O.Octree(sphere_mesh), O.Mesh(asource_xyz, grid["sphere_tri"]))
        self.c.set_anatomical(self.keep[0], self.keep[1], atarget_xyz, self.keep[2], grid["w_ptr"], grid["w_cp"], grid["w_val"], grid["face_ptr"], grid["face_idx"])

    def pairwise_table(self):
        return self.c.pairwise_table()

    def get_source_data(self):
        self.c.get_source_data()

    def unary_table(self):
        return self.c.unary_table(threads=ORACLE_THREADS)

    def triplet_table(self):
        return self.c.triplet_table()

    def triplet_octets(self, labeling, label):
        return self.c.triplet_octets(labeling, label, threads=ORACLE_THREADS)

    def total(self, labeling):
        return self.c.total(labeling)[0]


def angles(a, b):
    """angle (rad) between corresponding vertices of two spheres: the unit of north_star's 1e-4 rad bar"""
    ua = a / np.linalg.norm(a, axis=1, keepdims=True)
    ub = b / np.linalg.norm(b, axis=1, keepdims=True)
    return 2.0 * np.arcsin(np.minimum(1.0, 0.5 * np.linalg.norm(ua - ub, axis=1)))


def registration_parity(ctx, levels, D, **kw):
    """One multiresolution registration of the synthetic ico6 subject of bench.py (seeds 7 / 9) over the MI355X path and over the
    oracle, same caller loop, same optimiser: (max angle between the two registered spheres in rad, labelings identical?, number of
    labelings compared, seconds the oracle run took)."""
    import time

    import newmsm_amd as M
    from newmsm_amd import registration, synthetic

    xyz, tri = M.make_mesh_from_icosa(6)
    ref = synthetic.features(xyz, D, 7)
    src = synthetic.features(synthetic.known_warp(xyz, seed=9, rot_deg=3.0, amp=2.0), D, 7)
    lg, lw = [], []
    got = registration.run_multiresolution(registration.ProductOps(ctx), xyz, tri, src, xyz, tri, ref, levels, varnorm=True, labelings_out=lg, **kw)
    t0 = time.perf_counter()
    want = registration.run_multiresolution(OracleOps(M.mcmc_optimise), xyz, tri, src, xyz, tri, ref, levels, varnorm=True, labelings_out=lw, **kw)
    cpu_s = time.perf_counter() - t0
    same = len(lg) == len(lw) and all(np.array_equal(a, b) for a, b in zip(lg, lw))
    moved = float(angles(got[0], xyz).max())
    return dict(max_angle_rad=float(angles(got[0], want[0]).max()), labelings_identical=bool(same), labelings=len(lg), cpu_port_s=cpu_s,
                moved_rad=moved, energies_rel_diff=float(max(abs(a - b) / max(abs(b), 1e-300) for ea, eb in zip(got[2], want[2]) for a, b in zip(ea, eb))))
