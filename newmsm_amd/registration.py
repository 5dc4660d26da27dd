"""One resolution level of a pairwise discrete registration, as the reference's callers drive the hot path:

    Mesh_registration::run_discrete_opt                 M/mesh_registration.cpp:164-232
    NonLinearSRegDiscreteModel::Initialize              M/DiscreteModel.cpp:63-108
    NonLinearSRegDiscreteModel::setupCostFunction       M/DiscreteModel.cpp:216-262
    NonLinearSRegDiscreteModel::applyLabeling           M/DiscreteModel.cpp:264-269
    MCMC::optimise                                      M/mcmc_opt.h:31-134   (the optimiser without a licence restriction)

The loop itself is caller logic; everything it calls goes through an `ops` object.  `ProductOps` (below) maps the calls to
libmsmhip through the C ABI; the parity tests pass an oracle-backed object with the same methods, so that the two runs
differ in nothing but the implementation of the path (tests/helpers.py, tests/test_gpu_registration.py).
"""
import time

import numpy as np

from . import api

RAD = 100.0  # M/reg_tools.h


class ProductOps:
    """The calls of the level loop on the MI355X path (every method is one or two C-ABI calls)."""

    def __init__(self, ctx):
        self.ctx = ctx

    # --- meshes
    def icosphere(self, order):
        return api.make_mesh_from_icosa(order)

    def mesh(self, xyz, tri, feat=None):
        m = api.Mesh(self.ctx, xyz, tri)
        if feat is not None:
            m.set_pvalues(feat)
        return m

    def set_coords(self, mesh, xyz):
        mesh.set_coords(xyz)

    def coords(self, mesh):
        return mesh.get_coords()

    def unfold(self, mesh):
        return mesh.unfold(RAD)

    def sphere_project_warp(self, sphere_xyz, from_mesh, to_xyz):
        return api.sphere_project_warp(sphere_xyz, from_mesh, to_xyz)

    def warp_mesh(self, mesh, from_mesh, to_xyz):
        """sphere_project_warp of the coordinates `mesh` holds, in place (on the device: no host round trip of the 3 x V arrays)"""
        api.sphere_project_warp_mesh(mesh, from_mesh, to_xyz)

    # --- featurespace::initialise
    def metric_resample(self, in_mesh, data, new_mesh, slot=None):
        """slot (optional): a name under which the result may live in a pinned buffer of the context that the NEXT call with the same name reuses
        (the copy engine writes it directly: no staging memcpy of the D x V matrix); None: a fresh numpy array"""
        if slot is None:
            return api.metric_resample(in_mesh, data, new_mesh)
        out = self.ctx.scratch_host_array("metric_resample:" + slot, (np.atleast_2d(data).shape[0], new_mesh.V))
        return api.metric_resample(in_mesh, data, new_mesh, out=out)

    def pin(self, array):
        """a large input matrix in page-locked memory of the context for the length of a run (msm_host_alloc: one copy here, and every level's upload of
        it then skips the staging blocks).  Returns (token for unpin, the array to use); (None, array) when the array is small or not float64.
        (Until round 5 the caller's own numpy array was page-locked in place (msm_host_register).  Page-locking works on whole pages and the device address
        of such memory is its host address: the first and last page of an array in the heap are shared with its heap neighbours, and the HIP runtime
        page-locks and releases pageable buffers of asynchronous copies on its own -- whichever of two sharers is released first unmaps the page under the
        other.  msm_host_register now refuses ranges that are not whole pages)"""
        a = np.asarray(array)
        if a.dtype != np.float64 or a.nbytes < (1 << 20):
            return None, array
        try:
            pinned = self.ctx.host_array(a.shape)
        except api.MsmError:
            return None, array
        pinned[...] = a
        return pinned, pinned

    def unpin(self, token):
        if token is not None:
            self.ctx.release_host_array(token)

    def smooth_data(self, mesh, data, sigma):
        return api.smooth_data(mesh, data, mesh, sigma)

    def variance_normalise(self, data):
        return api.variance_normalise(data)

    def nearest_neighbour(self, mesh, data, q_xyz):
        return api.nearest_neighbour_interpolation(mesh, data, q_xyz)

    # --- aMSM (--regoption=5): Mesh_registration::resample_anatomy, M/mesh_registration.cpp:250-332
    def resample_anatomy_grid(self, cp_xyz, cp_tri, levels):
        return api.resample_anatomy_grid(cp_xyz, cp_tri, levels, RAD)

    def surface_resample(self, anat_xyz, sphere_mesh, q_xyz):
        """newresampler::surface_resample / project_anatomical_mesh (R/resampler.cpp:284-302, :260-282): the anatomy given on the vertices of sphere_mesh at q"""
        return api.barycentric_coords_resample(sphere_mesh, anat_xyz, q_xyz)

    # --- model host logic
    def cp_spacings(self, mesh, xyz, tri):
        return api.cp_spacings(xyz, tri)

    def estimate_triplets(self, mesh, tri):
        return api.estimate_triplets(tri)

    def estimate_pairs(self, mesh, tri, nodes):
        return api.estimate_pairs(tri, nodes)

    def label_sampling_grid(self, sg_order, max_dist):
        return api.label_sampling_grid(sg_order, max_dist)

    def rescale_sampling_grid(self, samples, scale):
        return api.rescale_sampling_grid(samples, scale)

    def cp_rotations(self, centre, cp_xyz):
        return api.cp_rotations(centre, cp_xyz)

    # --- cost function
    def cost(self, kind, simmeasure, rmode, params, target, source, cpgrid, src_feat):
        cf = api.DiscreteCostFunction(self.ctx, kind=kind, simmeasure=simmeasure, rmode=rmode, **params)
        cf.set_meshes(target, source, cpgrid)
        cf.set_featurespace(src_feat)
        return _ProductCost(cf)

    def mcmc(self, unary, tcosts, triplets, labeling, mcparam, iters, seed):
        return api.mcmc_optimise(unary, tcosts, triplets, labeling, mcparam=mcparam, iters=iters, seed=seed)

    def fusion_step(self, unary2, octets, triplets, passes, quads=None, pairs=None):
        return api.fusion_icm_step(unary2, octets, triplets, passes, quads=quads, pairs=pairs)

    def pairwise_solve(self, unary, paircosts, pairs, passes):
        return api.pairwise_icm(unary, paircosts, pairs, passes=passes)


class _ProductCost:
    def __init__(self, cf):
        self.cf = cf
        self._octets = None

    def reset_source(self, mesh):
        self.cf.reset_source(mesh)

    def reset_cpgrid(self, mesh):
        self.cf.reset_CPgrid(mesh)

    def set_spacings(self, maxsep, mvdmax):
        self.cf.set_spacings(maxsep, mvdmax)

    def set_cfweight(self, w):
        self.cf.set_dataaffintyweighting(w)

    def set_labels(self, labels, rot):
        self.cf.set_labels(labels, rot)

    def set_triplets(self, triplets):
        self.cf.setTriplets(triplets)

    def set_pairs(self, pairs):
        self.cf.setPairs(pairs)

    def set_anatomical(self, sphere_mesh, atarget_xyz, asource_xyz, grid):
        self.cf.set_anatomical(sphere_mesh, atarget_xyz, asource_xyz, grid["sphere_tri"], grid["w_ptr"], grid["w_cp"], grid["w_val"], grid["face_ptr"],
                               grid["face_idx"])

    def get_source_data(self):
        self.cf.get_source_data()

    def unary_table(self):
        return self.cf.computeUnaryCosts()

    def pairwise_table(self):
        return self.cf.computePairwiseCosts()

    def triplet_table(self):
        return self.cf.computeTripletCosts(pinned=True)  # consumed by the optimiser before the next table is computed

    def _octet_buffers(self):
        # the optimiser's per-step buffers: mapped pinned memory the kernel writes (two grow-only buffers per context, used in turn: a step's costs are
        # consumed by its solve while the next step may already be written into the other); the numpy views are made once per set of triplets
        if self._octets is None or self._octets[0].shape[0] != self.cf.T:
            self._octets = [self.cf.ctx.scratch_host_array("octets%d" % k, (self.cf.T, 8)) for k in range(2)]
            self._turn = 0
        return self._octets

    def triplet_octets(self, labeling, label):
        bufs = self._octet_buffers()
        out = bufs[self._turn]
        self._turn ^= 1
        return self.cf.tripletOctets(labeling, label, out)

    def prefetch_octets(self, labeling, label):
        """the NEXT triplet_octets call will ask for (labeling, label) unless the solve in between changes a label: let its kernel run meanwhile
        (msm_cost_triplet_octets_prefetch into the buffer that call will use)"""
        self.cf.prefetchTripletOctets(labeling, label, self._octet_buffers()[self._turn])

    def total(self, labeling):
        return self.cf.evaluateTotalCostSum(labeling)[0]


def hcp_msmall_levels(iterations=(10, 15, 15)):
    """The schedule of config/HCP_multimodal_alignment/MSMAllStrainFinalconf1to1_1to3_2 (BASELINE config 3) for run_multiresolution, read from
    the configuration text itself (newmsm_amd/config.py: PRESETS["HCP_MSMAll"] through the reference's grammar, float options rounded to
    float32 as the reference's option parser does): --CPgrid=2,3,4 --datagrid=4,5,6 --SGgrid=4,5,6 --it=10,15,15 --lambda=0.00001,0.0075,0.01
    --regoption=3 --regexp=2 --k_exponent=2 --bulkmod=1.6 --shearmod=0.4 --sigma_in/ref=0 --simval=2, with --triclique (the HO classes over
    the 32 feature rows of the MSMAll data), --rescaleL, --VN (pass varnorm=True) and --dopt=HOCR (optimiser "fusion": the label loop of
    Fusion::optimize; its binary solve is a stand-in, see run_discrete_level)."""
    from . import config

    return config.preset_levels("HCP_MSMAll", 32, iterations)[0]


def basic_levels(iterations=(3, 3, 3)):
    """Three DISCRETE levels shaped like config/basic_configs/config_standard_MSM_strain (--CPgrid=2,3,4 --datagrid=4,5,6 --SGgrid=4,5,6,
    --sigma_in/ref=4,2,1, pass varnorm=True for its --VN); optimiser, cost class and regulariser come from the caller's keyword arguments."""
    return [dict(data_order=4 + k, cp_order=2 + k, sigma_in=s, sigma_ref=s, iters=iterations[k]) for k, s in enumerate((4.0, 2.0, 1.0))]


def apply_labeling(rot, labels, labeling):
    """m_CPgrid.set_coord(i, m_ROT[i] * m_labels[labeling[i]]), operator*(Matrix, Point) R/point.cpp:207-213 (row sums left to right)"""
    R = np.asarray(rot).reshape(-1, 3, 3)
    v = np.asarray(labels)[np.asarray(labeling)]
    return R[:, :, 0] * v[:, None, 0] + R[:, :, 1] * v[:, None, 1] + R[:, :, 2] * v[:, None, 2]


def combine_costfunction_weighting(sourceweight, resampledtargetweight):
    """Mesh_registration::combine_costfunction_weighting, M/mesh_registration.cpp:849-869: the mean of the two weightings over
    the rows both have; the rows only the larger one has are kept"""
    a, b = np.atleast_2d(sourceweight), np.atleast_2d(resampledtargetweight)
    new = np.array(a if a.shape[0] >= b.shape[0] else b, dtype=np.float64)
    n = min(a.shape[0], b.shape[0])
    new[:n] = (a[:n] + b[:n]) / 2.0
    return new


def run_discrete_level(ops, target_xyz, target_tri, ref_feat, source_xyz, source_tri, src_feat, sph_reg, cp_order, *, sg_order=None,
                       iters=3, mciters=200, mcparam=0.8, seed=0, kind="univariate", simmeasure=2, rmode=3, labeldist=0.5,
                       rescale_labels=False, cost_params=None, timings=None, cp_start=None, in_weight=None, ref_weight=None,
                       optimiser="mcmc", icm_passes=5, converge=False, anat=None, speculate=True):
    """Runs `iters` iterations of run_discrete_opt for one level.  optimiser: "mcmc" -- the reference's Monte Carlo optimiser over the
    unary and T x L^3 triplet tables (M/mcmc_opt.h:31-134) -- or "fusion": the label loop of Fusion::optimize (I/Fusion/Fusion.h:136-229: two
    sweeps over the labels, per label step 2 N unary and 8 T triplet costs -- ONE fusion-move call on the MI355X path --, nodes that the
    binary solve gives 1 take the label), which is how every HCP configuration drives the hot path (--dopt=HOCR).  The binary solve itself
    (ELC + FastPD, licence-restricted) is replaced by a stand-in (ops.fusion_step: iterated conditional modes, icm_passes passes): such a run
    exercises and times the path as HOCR would, its result is not the reference's optimum.  "fastpd": --regoption=1 as --dopt=FastPD drives it
    (M/mesh_registration.cpp:182-188): the unary table and the P x L x L pair tables (computePairwiseCosts), then a stand-in for FPD::FastPD
    (ops.pairwise_solve: iterated conditional modes over all labels).  converge=True applies the reference's convergence test of a level
    (:204-213; the stand-in solves make its energies differ from the reference's, so it is off unless asked for).

    target / source: the reference and the moving sphere at the data resolution of this level (source_xyz = the sphere the
    moving features live on, sph_reg = its current registered position).  Returns (sph_reg, cp_xyz, energies, labelings).
    `timings` (optional dict) accumulates wall-clock seconds per phase.  cp_start: the control grid after warp_CPgrid (the
    warp of the previous level applied to it); the regular grid when None.  in_weight / ref_weight (rows x V, optional): the
    cost-function weightings of the moving and the reference data at this resolution (SPHin_CFWEIGHTING / SPHref_CFWEIGHTING);
    with both given every iteration resamples the reference weighting onto the moving sphere and averages the two
    (combine_weighting, M/mesh_registration.cpp:234-248), otherwise the weighting is all ones.
    speculate (fusion loop): queue the next label step's evaluations while the host solves the current one (ops' cost.prefetch_octets, if it has one).
    anat (rmode 4 / 5, aMSM): dict(order, in_anat, in_mesh, ref_anat, ref_mesh) -- --anatgrid of this level and the input / reference anatomical
    surfaces (V x 3) with the spheres (ops meshes) whose vertices they share (MESHES[0] / MESHES[1]).  initialize_level (M/mesh_registration.cpp:
    91-99) then prepares the anatomical regulariser: resample_anatomy (:250-332: the control grid retessellated to anatomical resolution with the
    face neighbourhoods -> NEARESTFACES, _ANATbaryweights, the input anatomy resampled onto it), the reference anatomy resampled the same way,
    set_anatomical_meshspace / set_anatomical_neighbourhood.  (setupCostFunction's reset_anatomical, M/DiscreteCostFunction.cpp:85-100, computes
    _aSOURCEtrans and MAXstrain, which nothing reads: not evaluated.)"""
    if sg_order is None:
        sg_order = cp_order + 2
    cost_params = dict(cost_params or {})
    clock = timings if timings is not None else {}

    def timed(name, fn, *a):
        t0 = time.perf_counter()
        out = fn(*a)
        clock[name] = clock.get(name, 0.0) + time.perf_counter() - t0
        return out

    # --- initialize_level / Initialize(CONTROL)
    cp_xyz, cp_tri = ops.icosphere(cp_order)
    target = ops.mesh(target_xyz, target_tri, ref_feat)
    source = ops.mesh(source_xyz, source_tri)
    cpgrid = ops.mesh(cp_xyz, cp_tri)
    maxsep, mvdmax = ops.cp_spacings(cpgrid, cp_xyz, cp_tri)
    samples, barycentres = ops.label_sampling_grid(sg_order, labeldist * mvdmax)
    centre = samples[0]
    pairwise = optimiser == "fastpd"  # --regoption=1: the model lists pairs instead of triplets (M/DiscreteModel.cpp:40,98-101,258-259)
    if pairwise and rmode != 1:
        raise ValueError("MeshREG ERROR:: you cannot run higher order clique regularisers with fastPD ")  # M/mesh_registration.cpp:759-760
    triplets = None if pairwise else ops.estimate_triplets(cpgrid, cp_tri)
    pairs = ops.estimate_pairs(cpgrid, cp_tri, len(cp_xyz)) if pairwise else None
    cost = ops.cost(kind, simmeasure, rmode, cost_params, target, source, cpgrid, src_feat)  # set_meshes: _ORIG, _oCPgrid
    cost.set_spacings(maxsep, mvdmax)
    if rmode in (4, 5):
        if anat is None:  # M/mesh_registration.cpp:100-104
            raise ValueError("--regoption 4 has been removed from newMSM. Use --regoption 3 for spherical mesh regularisation or --regoption 5 for anatomical mesh "
                             "regularisation." if rmode == 4 else
                             "--regoption 5 requires anatomical meshes. Use --regoption 3 for spherical mesh regularisation or provide anatomical meshes.")
        if pairwise:
            raise ValueError("MeshREG ERROR:: you cannot run higher order clique regularisers with fastPD ")
        grid = ops.resample_anatomy_grid(cp_xyz, cp_tri, max(0, anat["order"] - cp_order))     # ANAT_ico, _ANATbaryweights, NEARESTFACES
        anat_orig = timed("surface_resample", ops.surface_resample, anat["in_anat"], anat["in_mesh"], grid["sphere_xyz"])    # ANAT_orig (:323)
        anat_target = timed("surface_resample", ops.surface_resample, anat["ref_anat"], anat["ref_mesh"], grid["sphere_xyz"])  # ANAT_target (:93)
        cost.set_triplets(triplets)  # NEARESTFACES is indexed by triplet = control triangle
        cost.set_anatomical(ops.mesh(grid["sphere_xyz"], grid["sphere_tri"]), anat_target, anat_orig, grid)
    m_iter, m_scale = 1, 1.0
    energies, labelings = [], []
    energy = 0.0
    sph_reg = np.array(sph_reg, dtype=np.float64)
    if cp_start is not None:
        cp_xyz = np.array(cp_start, dtype=np.float64)
    for it in range(iters):
        # --- reset_meshspace + setupCostFunction
        ops.set_coords(source, sph_reg)
        if in_weight is not None and ref_weight is not None:  # setupCostFunctionWeighting(combine_weighting())
            resampled = timed("metric_resample", ops.metric_resample, target, ref_weight, source)
            cost.set_cfweight(combine_costfunction_weighting(in_weight, resampled))
        cost.reset_source(source)
        ops.set_coords(cpgrid, cp_xyz)
        cost.reset_cpgrid(cpgrid)
        rot = ops.cp_rotations(centre, cp_xyz)
        if rescale_labels:
            labels, m_scale = ops.rescale_sampling_grid(samples, m_scale)
        elif m_iter % 2 == 0:
            labels = samples
        else:
            labels = barycentres
        cost.set_labels(labels, rot)
        timed("get_source_data", cost.get_source_data)
        if pairwise:
            cost.set_pairs(pairs)
        else:
            cost.set_triplets(triplets)
        m_iter += 1
        unary = timed("unary_table", cost.unary_table)
        labeling = np.zeros(len(cp_xyz), dtype=np.int32)  # resetLabeling
        if optimiser == "fusion":  # --- Fusion::optimize: computeUnaryCosts, then per label step the 8 T combinations
            nodes = np.arange(len(cp_xyz))
            L = len(labels)
            steps = [(sweep, label) for sweep in range(2) for label in range(L)]
            prefetch = getattr(cost, "prefetch_octets", None) if speculate else None
            for k, (sweep, label) in enumerate(steps):
                if not np.any(labeling != label):
                    continue
                octets = timed("fusion_moves", cost.triplet_octets, labeling, label)
                if prefetch is not None:
                    # While the host solves this step the GPU would idle; most steps of a converging level change no label, and then the next step's
                    # evaluations are a function of what is known now: queue them (a hint: the results do not depend on it, a step whose labeling did
                    # change is evaluated afresh)
                    nxt = next((lb for _, lb in steps[k + 1:] if np.any(labeling != lb)), None)
                    if nxt is not None:
                        timed("fusion_prefetch", prefetch, labeling, nxt)
                unary2 = np.stack([unary[labeling, nodes], unary[label]], axis=1)
                x = timed("optimiser", ops.fusion_step, unary2, octets, triplets, icm_passes)
                labeling = np.where((x == 1) & (labeling != label), label, labeling).astype(np.int32)
        elif pairwise:  # --- FastPD: computeUnaryCosts, computePairwiseCosts, FPD::FastPD(model, 100) (M/mesh_registration.cpp:182-188)
            paircosts = timed("pairwise_table", cost.pairwise_table)
            labeling = timed("optimiser", ops.pairwise_solve, unary, paircosts, pairs, 100)
        else:  # --- MCMC: computeUnaryCosts, computeTripletCosts, optimise
            tcosts = timed("triplet_table", cost.triplet_table)
            labeling = timed("optimiser", ops.mcmc, unary, tcosts, triplets, labeling, mcparam, mciters, seed + it)
        newenergy = timed("total_cost", cost.total, labeling)
        # the level's convergence test (M/mesh_registration.cpp:204-213): from the fourth iteration on, every second one, not for MCMC; the
        # iteration that triggers it is not applied
        if converge and it > 2 and (it - 1) % 2 == 0 and energy - newenergy < 0.001 and optimiser != "mcmc":
            break
        energy = newenergy
        energies.append(newenergy)
        labelings.append(labeling)
        # --- applyLabeling, warp the source through the control grid move, unfold both
        new_cp = apply_labeling(rot, labels, labeling)
        # (the source mesh holds sph_reg since the top of the iteration: it is warped where it lies; cpgrid still holds the previous grid)
        timed("sphere_project_warp", ops.warp_mesh, source, cpgrid, new_cp)
        ops.set_coords(cpgrid, new_cp)
        timed("unfold", ops.unfold, cpgrid)
        cp_xyz = ops.coords(cpgrid)
        timed("unfold", ops.unfold, source)
        sph_reg = ops.coords(source)
    return sph_reg, cp_xyz, energies, labelings


def run_multiresolution(ops, in_xyz, in_tri, in_data, ref_xyz, ref_tri, ref_data, levels, *, varnorm=False, timings=None, in_cfweight=None,
                        ref_cfweight=None, labelings_out=None, in_anat=None, ref_anat=None, **level_kw):
    """Mesh_registration::run_multiresolutions (M/mesh_registration.cpp:30-50) for DISCRETE levels without file I/O:

    per level  featurespace::initialise (M/featurespace.cpp:39-86: metric_resample of both data sets onto the level's
               icosphere, smooth_data, variance_normalise), project_CPgrid (M/mesh_registration.cpp:131-162: the warp of the
               previous level carried to the new data grid and control grid, unfold) and run_discrete_opt;
    at the end transform (:352-356): the input sphere moved through the final warp ("sphere.reg").

    in_* / ref_*: the input and reference spheres (radius 100) with their D x V data.  levels: dicts with data_order, cp_order
    and optionally sg_order, sigma_in, sigma_ref, iters, mciters, cost_params.  in_cfweight / ref_cfweight (rows x V on the
    input / reference sphere, optional): cost-function weightings, brought to each level's grid by nearest-neighbour
    interpolation (downsample_cfweighting, M/mesh_registration.cpp:334-350).  labelings_out (optional list): receives every iteration's labeling, level
    after level (the parity tests compare the optimiser's decisions of two runs).  in_anat / ref_anat (V x 3 on the vertices of the input /
    reference sphere; both or none, CLI/newmsm.cpp:40-45): the anatomical surfaces of a --regoption=5 (aMSM) run; a level's "anat_order" is its
    --anatgrid.  recentre() of the regular spheres
    (a shift of ~1e-15) is not applied.  Returns (sphere_reg, per-level registered data grids, per-level energies)."""
    clock = timings if timings is not None else {}

    def timed(name, fn, *a):
        t0 = time.perf_counter()
        out = fn(*a)
        clock[name] = clock.get(name, 0.0) + time.perf_counter() - t0
        return out

    in_xyz = np.asarray(in_xyz, dtype=np.float64)
    in_mesh, ref_mesh = ops.mesh(in_xyz, in_tri), ops.mesh(ref_xyz, ref_tri)
    sph_reg_prev, prev_order, regs, all_energies = None, None, [], []
    # the two data matrices go up once per level: in page-locked memory for the run, their uploads skip the staging blocks (ProductOps.pin; absent elsewhere)
    pins = []
    try:
        if hasattr(ops, "pin"):
            for which in (0, 1):
                token, arr = ops.pin(in_data if which == 0 else ref_data)
                pins.append(token)
                if which == 0:
                    in_data = arr
                else:
                    ref_data = arr
        return _run_levels(ops, clock, timed, in_xyz, in_mesh, ref_mesh, in_data, ref_data, levels, varnorm, in_cfweight, ref_cfweight, labelings_out, in_anat,
                           ref_anat, ref_xyz, level_kw, sph_reg_prev, prev_order, regs, all_energies)
    finally:
        for t in pins:
            ops.unpin(t)


def _run_levels(ops, clock, timed, in_xyz, in_mesh, ref_mesh, in_data, ref_data, levels, varnorm, in_cfweight, ref_cfweight, labelings_out, in_anat, ref_anat,
                ref_xyz, level_kw, sph_reg_prev, prev_order, regs, all_energies):
    """the level loop of run_multiresolution (see there)"""
    for lv in levels:
        ico_xyz, ico_tri = ops.icosphere(lv["data_order"])
        ico = ops.mesh(ico_xyz, ico_tri)
        feats = []
        for mesh, data, sigma, slot in ((in_mesh, in_data, lv.get("sigma_in", 0.0), "in"), (ref_mesh, ref_data, lv.get("sigma_ref", 0.0), "ref")):
            # (a level's matrices are consumed -- uploaded by the cost function and the target mesh -- before the next level asks for its own: the
            # result slots are reused from level to level)
            f = timed("metric_resample", ops.metric_resample, mesh, data, ico, slot)
            if sigma > 0.0:
                f = timed("smooth_data", ops.smooth_data, ico, f, sigma)
            if varnorm:
                f = ops.variance_normalise(f)
            feats.append(f)
        cp_start = None
        if sph_reg_prev is None:
            sph_in = ico_xyz  # level 1, no transformed mesh: project_CPgrid only unfolds the (regular) data grid
        else:
            prev_xyz, prev_tri = ops.icosphere(prev_order)
            incurrent = timed("sphere_project_warp", ops.sphere_project_warp, in_xyz, ops.mesh(prev_xyz, prev_tri), sph_reg_prev)
            sph_in = timed("sphere_project_warp", ops.sphere_project_warp, ico_xyz, in_mesh, incurrent)
            cp_xyz, cp_tri = ops.icosphere(lv["cp_order"])
            cpm = ops.mesh(timed("sphere_project_warp", ops.sphere_project_warp, cp_xyz, in_mesh, incurrent), cp_tri)  # warp_CPgrid
            timed("unfold", ops.unfold, cpm)
            cp_start = ops.coords(cpm)
        moved = ops.mesh(sph_in, ico_tri)
        timed("unfold", ops.unfold, moved)
        sph_in = ops.coords(moved)
        kw = dict(level_kw)
        kw.update({k: lv[k] for k in ("sg_order", "iters", "mciters", "mcparam", "cost_params", "kind", "rescale_labels", "optimiser", "simmeasure", "rmode",
                                      "converge") if k in lv})
        if in_anat is not None or ref_anat is not None:
            if in_anat is None or ref_anat is None:
                raise ValueError("Error: must supply both anatomical meshes or none")  # CLI/newmsm.cpp:41-43
            if len(in_anat) != len(in_xyz) or len(ref_anat) != len(np.asarray(ref_xyz)):
                raise ValueError("MeshREG ERROR:: input/reference anatomical mesh resolution is inconsistent with input/reference spherical mesh resolution.")
            kw["anat"] = dict(order=lv.get("anat_order", lv["cp_order"] + 2), in_anat=np.asarray(in_anat, dtype=np.float64), in_mesh=in_mesh,
                              ref_anat=np.asarray(ref_anat, dtype=np.float64), ref_mesh=ref_mesh)
        if in_cfweight is not None and ref_cfweight is not None:
            kw["in_weight"] = ops.nearest_neighbour(in_mesh, in_cfweight, ico_xyz)
            kw["ref_weight"] = ops.nearest_neighbour(ref_mesh, ref_cfweight, ico_xyz)
        sph_reg, _, energies, labelings = run_discrete_level(ops, ico_xyz, ico_tri, feats[1], ico_xyz, ico_tri, feats[0], sph_in, lv["cp_order"],
                                                             cp_start=cp_start, timings=clock, **kw)
        if labelings_out is not None:
            labelings_out.extend(labelings)
        regs.append(sph_reg)
        all_energies.append(energies)
        sph_reg_prev, prev_order = sph_reg, lv["data_order"]
    last_xyz, last_tri = ops.icosphere(levels[-1]["data_order"])
    sphere_reg = timed("sphere_project_warp", ops.sphere_project_warp, in_xyz, ops.mesh(last_xyz, last_tri), sph_reg_prev)
    return sphere_reg, regs, all_energies
