"""Assembles one discrete-optimisation iteration of a pairwise registration (what
NonLinearSRegDiscreteModel::Initialize + setupCostFunction prepare, M/DiscreteModel.cpp:63-108, :216-262)
from synthetic inputs, using only libmsmhip's [host] entry points.  Returns plain numpy arrays so that
tests can feed the very same numbers to the oracle.
"""
import numpy as np

from . import api, synthetic


def pairwise_inputs(data_order=6, cp_order=4, sg_order=None, D=1, seed=1234, warp_amp=0.6, warp_rot=2.0, rescale=True,
                    labeldist=0.5, target_noise=0.0, target_warp=0.0, target_radial=0.0):
    """Inputs of one iteration: target / source / control grids, features, labels, rotations, cliques."""
    if sg_order is None:
        sg_order = cp_order + 2
    txyz0, ttri = api.make_mesh_from_icosa(data_order)  # reference sphere (regular unless asked otherwise)
    txyz = txyz0
    if target_warp > 0:  # smooth deformation: irregular triangles, still a simple (fold-free) surface
        txyz = synthetic.known_warp(txyz, seed=seed + 5, rot_deg=0.0, amp=target_warp)
    if target_noise > 0:  # vertex jitter: slivers and folds -> several triangles can contain a projection
        rng = np.random.default_rng(seed + 6)
        txyz = txyz + rng.normal(scale=target_noise, size=txyz.shape)
        txyz = txyz / np.linalg.norm(txyz, axis=1, keepdims=True) * synthetic.RAD
    if target_radial > 0:  # star-shaped but not spherical: vertices move radially (still a simple surface; off the query shell)
        txyz = txyz * (1.0 + target_radial * synthetic.smooth_feature(txyz, 1, seed + 7))[:, None]
    cxyz, ctri = api.make_mesh_from_icosa(cp_order)    # control grid
    # source: the data sphere part-way through a registration (smoothly warped), features of the moving image
    sxyz = synthetic.known_warp(txyz0, seed=seed + 1, rot_deg=warp_rot, amp=warp_amp)
    ref_feat = synthetic.features(txyz, D, seed)
    src_feat = synthetic.features(synthetic.known_warp(txyz0, seed=seed + 2, rot_deg=1.5 * warp_rot, amp=2 * warp_amp), D, seed)
    # the control grid rides along with the source (warp_CPgrid): move it through the same warp
    cxyz_cur = synthetic.known_warp(cxyz, seed=seed + 1, rot_deg=warp_rot, amp=warp_amp)
    maxsep, mvdmax = api.cp_spacings(cxyz_cur, ctri)
    samples, barycentres = api.label_sampling_grid(sg_order, labeldist * mvdmax)
    if rescale:
        labels, _ = api.rescale_sampling_grid(samples, 1.0)
    else:
        labels = samples
    rot = api.cp_rotations(samples[0], cxyz_cur)
    return dict(data_order=data_order, cp_order=cp_order, sg_order=sg_order, D=D,
                target_xyz=txyz, target_tri=ttri, source_xyz=sxyz, source_tri=ttri, source_orig_xyz=txyz0,
                cp_xyz=cxyz_cur, cp_orig_xyz=cxyz, cp_tri=ctri, ref_feat=ref_feat, src_feat=src_feat,
                maxsep=maxsep, mvdmax=mvdmax, samples=samples, barycentres=barycentres, labels=labels, rot=rot,
                triplets=api.estimate_triplets(ctri), pairs=api.estimate_pairs(ctri, len(cxyz)))


def build_cost(ctx, inp, kind="univariate", simmeasure=2, rmode=3, **params):
    """Product-side objects for `inp` (set_meshes happens with the ORIGINAL source / control grid, as in
    Initialize(); the current coordinates arrive through reset_source / reset_CPgrid)."""
    target = api.Mesh(ctx, inp["target_xyz"], inp["target_tri"])
    target.set_pvalues(inp["ref_feat"])
    source = api.Mesh(ctx, inp["source_orig_xyz"], inp["source_tri"])
    cpgrid = api.Mesh(ctx, inp["cp_orig_xyz"], inp["cp_tri"])
    cf = api.DiscreteCostFunction(ctx, kind=kind, simmeasure=simmeasure, rmode=rmode, **params)
    cf.set_meshes(target, source, cpgrid)
    source.set_coords(inp["source_xyz"])
    cpgrid.set_coords(inp["cp_xyz"])
    cf.reset_source(source)
    cf.reset_CPgrid(cpgrid)
    cf.set_featurespace(inp["src_feat"])
    cf.set_spacings(inp["maxsep"], inp["mvdmax"])
    cf.set_labels(inp["labels"], inp["rot"])
    if rmode == 1:
        cf.setPairs(inp["pairs"])
    else:
        cf.setTriplets(inp["triplets"])
    return cf, dict(target=target, source=source, cpgrid=cpgrid)


def anatomical_inputs(ctx, inp, anat_order=None, seed=99):
    """Inputs of the anatomical regulariser (regoption 5) as Mesh_registration::resample_anatomy (M/mesh_registration.cpp:250-332) prepares them --
    api.resample_anatomy_grid: the control grid retessellated to anatomical resolution with the reference's face-neighbourhood bookkeeping
    (NEARESTFACES in its order, _ANATbaryweights in its overwrite order) -- with a synthetic source and target anatomy on the vertices of that
    sphere (what surface_resample of the two anatomical surfaces would deliver)."""
    if anat_order is None:
        anat_order = inp["cp_order"] + 2
    grid = api.resample_anatomy_grid(inp["cp_orig_xyz"], inp["cp_tri"], anat_order - inp["cp_order"], synthetic.RAD)
    axyz = grid["sphere_xyz"]
    # two smooth "cortical" surfaces on the aICO vertices
    d = axyz / synthetic.RAD
    rs = 60.0 + 6.0 * synthetic.smooth_feature(axyz, 0, seed) + 3.0 * synthetic.smooth_feature(axyz, 1, seed)
    rt = 62.0 + 5.0 * synthetic.smooth_feature(axyz, 2, seed + 1) + 4.0 * synthetic.smooth_feature(axyz, 0, seed + 2)
    return dict(grid, anat_order=anat_order, asource_xyz=d * rs[:, None], atarget_xyz=d * rt[:, None])


def build_group(ctx, S, data_order=6, cp_order=4, D=2, subjects=None, lambda_=0.2, simmeasure=2, seed=40, template_order=None, label_order_offset=2):
    """A synthetic groupwise (gMSM) problem part-way through a registration, BASELINE config 5 shape: S subjects on the data grid
    ico<data_order>, each with its own smooth warp so far and D feature rows, the template = the regular sphere at data
    resolution, control grid ico<cp_order>, the unrescaled sampling-grid labels (DiscreteGroupModel::setupCostFunction
    M/DiscreteGroupModel.cpp:163-196).  `subjects` (optional): the subjects whose data this process holds (a shard); the others
    are registered with their control grids only and must be imported.  Returns (group, keep-alive list)."""
    dxyz, dtri = api.make_mesh_from_icosa(data_order)
    cxyz, ctri = api.make_mesh_from_icosa(cp_order)
    _, mvd = api.cp_spacings(cxyz, ctri)
    samples, _ = api.label_sampling_grid(cp_order + label_order_offset, 0.5 * mvd)  # (offset 2: the 19 labels of the reference's sampling grid; 3, 4: 61, 217 -- tests)
    g = api.DiscreteGroupCostFunction(ctx, S, simmeasure=simmeasure, lambda_=lambda_)
    txyz, ttri = dxyz, dtri
    if template_order == "morton":  # the same template with its vertices renumbered along a space-filling curve
        q = np.clip(((dxyz + 101.0) / 202.0 * 1024.0).astype(np.int64), 0, 1023)
        code = np.zeros(len(dxyz), dtype=np.int64)
        for b in range(10):
            for a in range(3):
                code |= ((q[:, a] >> b) & 1) << (3 * b + 2 - a)
        perm = np.argsort(code, kind="stable")
        rank = np.empty_like(perm)
        rank[perm] = np.arange(len(perm))
        txyz, ttri = dxyz[perm], rank[dtri].astype(np.int32)
    tm = api.Mesh(ctx, txyz, ttri)
    g.set_template(tm, None)
    g.Initialize(cxyz, ctri)
    keep = [tm]
    mine = range(S) if subjects is None else subjects
    for s in range(S):
        if s in mine:
            feat = synthetic.features(synthetic.known_warp(dxyz, seed=seed + 50 + s, rot_deg=2.0, amp=1.0), D, seed=5)
            regular = api.Mesh(ctx, dxyz, dtri)
            g.reset_meshspace(s, regular, feat)   # first call: _ORIG_MESHES = the regular sphere
            regular.set_coords(synthetic.known_warp(dxyz, seed=seed + s, rot_deg=1.0 + 0.1 * s, amp=0.5))
            g.reset_meshspace(s, regular, feat)
            keep.append(regular)
        g.reset_CPgrid(s, synthetic.known_warp(cxyz, seed=seed + s, rot_deg=1.0 + 0.1 * s, amp=0.5))
    g.set_labels(samples)
    return g, keep
