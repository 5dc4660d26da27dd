"""End to end over one resolution level: the caller loop of run_discrete_opt (newmsm_amd/registration.py) drives the MI355X
path and the oracle with the same inputs, the same optimiser and the same seed.  BASELINE.json's north star asks for
registered vertex coordinates within 1e-4 rad of the CPU run; the path is built to the reference's operation order, so the
runs agree far more closely (the tables differ by ~1e-15 where the GPU sums in parallel; everything else is bit-exact)."""
import numpy as np
import pytest

import newmsm_amd as M
from newmsm_amd import registration, synthetic

from helpers import OracleOps

pytestmark = pytest.mark.gpu
NORTH_STAR_TOL_RAD = 1e-4


from helpers import angles  # noqa: E402


def level_inputs(data_order, D, seed):
    xyz, tri = M.make_mesh_from_icosa(data_order)
    ref = synthetic.features(xyz, D, seed)
    src = synthetic.features(synthetic.known_warp(xyz, seed=seed + 2, rot_deg=4.0, amp=2.5), D, seed)  # the same pattern, displaced
    return xyz, tri, ref, src


@pytest.mark.parametrize("kind,D,rescale", [("univariate", 1, False), ("multivariate", 3, True)])
def test_level_matches_oracle(ctx, kind, D, rescale):
    xyz, tri, ref, src = level_inputs(4, D, seed=21)
    kw = dict(cp_order=2, iters=3, mciters=60, mcparam=0.3, seed=5, kind=kind, rescale_labels=rescale, cost_params=dict(lambda_=0.05))
    got = registration.run_discrete_level(registration.ProductOps(ctx), xyz, tri, ref, xyz, tri, src, xyz, **kw)
    want = registration.run_discrete_level(OracleOps(M.mcmc_optimise), xyz, tri, ref, xyz, tri, src, xyz, **kw)
    for a, b in zip(got[3], want[3]):
        assert np.array_equal(a, b)  # the optimiser took the same decisions from both sets of tables
    assert np.allclose(got[2], want[2], rtol=1e-10)
    moved = angles(got[0], xyz)
    assert moved.max() > 1e-3 and len({tuple(l) for l in got[3]}) > 1  # the registration did move the source
    assert angles(got[0], want[0]).max() <= NORTH_STAR_TOL_RAD
    assert np.abs(got[0] - want[0]).max() < 1e-9 and np.abs(got[1] - want[1]).max() < 1e-9


def test_level_with_cost_function_weightings(ctx):
    """--inweight / --refweight: every iteration resamples the reference weighting onto the moving sphere and combines the two"""
    xyz, tri, ref, src = level_inputs(4, 3, seed=41)
    w_in = 0.5 + 0.5 * np.abs(synthetic.features(xyz, 1, 43))
    w_ref = 0.5 + 0.5 * np.abs(synthetic.features(xyz, 3, 44))  # per-dimension weighting of the reference
    kw = dict(cp_order=2, iters=2, mciters=40, mcparam=0.3, seed=6, kind="multivariate", cost_params=dict(lambda_=0.05), in_weight=w_in, ref_weight=w_ref)
    got = registration.run_discrete_level(registration.ProductOps(ctx), xyz, tri, ref, xyz, tri, src, xyz, **kw)
    want = registration.run_discrete_level(OracleOps(M.mcmc_optimise), xyz, tri, ref, xyz, tri, src, xyz, **kw)
    plain = registration.run_discrete_level(registration.ProductOps(ctx), xyz, tri, ref, xyz, tri, src, xyz, **dict(kw, in_weight=None, ref_weight=None))
    for a, b in zip(got[3], want[3]):
        assert np.array_equal(a, b)
    assert np.allclose(got[2], want[2], rtol=1e-10) and np.abs(got[0] - want[0]).max() < 1e-9
    assert not np.allclose(got[2], plain[2], rtol=1e-6)  # the weighting does change the energies


def test_level_improves_the_similarity(ctx):
    """sanity of the assembled loop: a few iterations bring a displaced copy of the pattern closer to the reference"""
    xyz, tri, ref, src = level_inputs(5, 1, seed=3)
    reg, _, energies, _ = registration.run_discrete_level(registration.ProductOps(ctx), xyz, tri, ref, xyz, tri, src, xyz, cp_order=3, iters=4,
                                                          mciters=100, mcparam=0.3, seed=1, cost_params=dict(lambda_=0.01))
    tm, sm = M.Mesh(ctx, xyz, tri), M.Mesh(ctx, reg, tri)
    before = np.corrcoef(ref[0], src[0])[0, 1]
    after = np.corrcoef(ref[0], M.metric_resample(sm, src, tm)[0])[0, 1]
    assert 1.0 - after < 0.6 * (1.0 - before), (before, after)  # the smooth synthetic pattern starts at r = 0.99


def test_two_levels_match_oracle(ctx):
    """run_multiresolutions over two DISCRETE levels: resample + smooth + normalise the data per level, carry the warp of level 1
    to the data grid and control grid of level 2 (project_CPgrid), register, and move the input sphere through the result"""
    in_xyz, in_tri = M.make_mesh_from_icosa(5)
    ref_xyz = in_xyz
    ref = synthetic.features(ref_xyz, 2, 31)
    src = synthetic.features(synthetic.known_warp(in_xyz, seed=33, rot_deg=4.0, amp=2.5), 2, 31)
    levels = [dict(data_order=3, cp_order=1, sigma_in=4.0, sigma_ref=4.0, iters=2, mciters=40),
              dict(data_order=4, cp_order=2, sigma_in=2.0, sigma_ref=0.0, iters=2, mciters=40)]
    kw = dict(varnorm=True, mcparam=0.3, seed=9, kind="multivariate", cost_params=dict(lambda_=0.05),
              in_cfweight=0.5 + 0.5 * np.abs(synthetic.features(in_xyz, 1, 35)), ref_cfweight=0.5 + 0.5 * np.abs(synthetic.features(ref_xyz, 1, 36)))
    got = registration.run_multiresolution(registration.ProductOps(ctx), in_xyz, in_tri, src, ref_xyz, in_tri, ref, levels, **kw)
    want = registration.run_multiresolution(OracleOps(M.mcmc_optimise), in_xyz, in_tri, src, ref_xyz, in_tri, ref, levels, **kw)
    assert np.allclose(np.concatenate(got[2]), np.concatenate(want[2]), rtol=1e-9)
    for a, b in zip(got[1], want[1]):
        assert np.abs(a - b).max() < 1e-9
    assert angles(got[0], want[0]).max() <= NORTH_STAR_TOL_RAD and np.abs(got[0] - want[0]).max() < 1e-9
    assert angles(got[0], in_xyz).max() > 1e-3


@pytest.mark.parametrize("fmt", ["GIFTI", "ASCII"])
def test_files_in_files_out(ctx, tmp_path, fmt):
    """tools/register_files.py under newmsm's flag names (CLI/msmOptions.h:59-157: --inmesh --refmesh --indata --refdata --conf --out -f) with the
    shipped standard_MSM_strain configuration (config/basic_configs/config_standard_MSM_strain; its AFFINE level is reported as skipped, fewer
    iterations for the test's sake): the three outputs of run_multiresolutions (M/mesh_registration.cpp:47-49) appear under the reference's names --
    <out>sphere.reg, <out>sphere.LR.reg (M/mesh_registration.h:170), <out>transformed_and_reprojected -- and hold what the same schedule gives when
    run_multiresolution is called directly (to the float32 of the files)."""
    import os
    import subprocess
    import sys

    from newmsm_amd import config, meshio

    xyz, tri = M.make_mesh_from_icosa(5)
    ref = synthetic.features(xyz, 1, 5)
    src = synthetic.features(synthetic.known_warp(xyz, seed=8, rot_deg=3.0, amp=2.0), 1, 5)
    d = str(tmp_path) + "/"
    text = config.PRESETS["standard_MSM_strain"].replace("--it=50,20,25,25", "--it=50,2,2,2").replace("--datagrid=5,5,5,6", "--datagrid=5,4,5,5").replace("--SGgrid=0,4,5,6", "--SGgrid=0,4,5,5").replace("--CPgrid=0,2,3,4", "--CPgrid=0,2,3,3")
    assert "--opt=AFFINE,DISCRETE,DISCRETE,DISCRETE" in text and "--it=50,2,2,2" in text
    with open(d + "conf", "w") as f:
        f.write(text)
    if fmt == "GIFTI":
        files = dict(mesh=d + "in.surf.gii", indata=d + "in.func.gii", refdata=d + "ref.func.gii")
        meshio.save_surface(files["mesh"], xyz, tri)
        meshio.save_metric(files["indata"], src)
        meshio.save_metric(files["refdata"], ref)
        surf, data = ".surf.gii", ".func.gii"
    else:
        files = dict(mesh=d + "in.asc", indata=d + "in_data.asc", refdata=d + "ref_data.asc")
        meshio.save_ascii(files["mesh"], xyz, tri)
        meshio.save_ascii(files["indata"], xyz, tri, src[0])
        meshio.save_ascii(files["refdata"], xyz, tri, ref[0])
        surf, data = ".asc", ".dpv"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, "tools/register_files.py", "--inmesh=" + files["mesh"], "--refmesh=" + files["mesh"], "--indata=" + files["indata"],
                          "--refdata=" + files["refdata"], "--conf=" + d + "conf", "--out=" + d + "out.", "-f", fmt, "--verbose"],
                         cwd=root, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr
    assert "level 1 (--opt=AFFINE)" in run.stderr and "skipped" in run.stderr
    reg, rtri = meshio.load_surface(d + "out.sphere.reg" + surf)
    lr, lrtri = meshio.load_surface(d + "out.sphere.LR.reg" + surf)
    assert np.array_equal(rtri, tri) and np.allclose(np.linalg.norm(reg, axis=1), 100.0, atol=1e-3) and angles(reg, xyz).max() > 1e-4
    assert np.array_equal(lrtri, tri) and np.allclose(np.linalg.norm(lr, axis=1), 100.0, atol=1e-3)   # the last level's data grid is ico5 as well
    moved = meshio.load_data(d + "out.transformed_and_reprojected" + data, len(xyz))
    assert moved.shape == (1, len(xyz))
    # the same schedule run directly on what the files hold (float32 data, the surface's coordinates through a float32 / a text file)
    in_xyz, _ = meshio.load_surface(files["mesh"])
    in_xyz = in_xyz - in_xyz.mean(axis=0)
    in_xyz = in_xyz * (100.0 / np.linalg.norm(in_xyz, axis=1, keepdims=True))
    src_f, ref_f = meshio.load_data(files["indata"], len(xyz)), meshio.load_data(files["refdata"], len(xyz))
    levels, run_kw, skipped = config.levels_from_config(config.parse_config(text), 1)
    want, regs, _ = registration.run_multiresolution(registration.ProductOps(ctx), in_xyz, tri, src_f, in_xyz, tri, ref_f, levels, **run_kw)
    assert skipped == [(0, "AFFINE")]
    assert np.abs(reg - want).max() < (2e-5 if fmt == "GIFTI" else 1e-3) and np.abs(lr - regs[-1]).max() < (2e-5 if fmt == "GIFTI" else 1e-3)
    direct = M.metric_resample(M.Mesh(ctx, want, tri), src_f, M.Mesh(ctx, in_xyz, tri))
    assert np.abs(moved - direct).max() < 1e-3 * max(1.0, np.abs(direct).max())
    # the closer the better: the moved data correlate with the reference better than the unmoved ones
    assert np.corrcoef(moved[0], ref[0])[0, 1] > np.corrcoef(src[0], ref[0])[0, 1]


def test_groupwise_files_in_files_out(ctx, tmp_path):
    """tools/register_files.py --groupwise under newmsm's flag names (CLI/msmOptions.h:73-87: -g --meshes --template --data --mask): mesh and data lists
    as text files (read_ascii_list), two levels, three subjects; the outputs of Group_Mesh_registration (M/group_mesh_registration.cpp:120-133,
    .h:79-82) appear per subject -- <out>sphere-<i>.reg, <out>sphere-<i>.LR.reg, <out>transformed_and_reprojected-<i> -- and hold what
    run_group_multiresolution gives on what the files hold."""
    import os
    import subprocess
    import sys

    from newmsm_amd import config, group_registration, meshio

    S = 3
    xyz, tri = M.make_mesh_from_icosa(4)
    d = str(tmp_path) + "/"
    text = "--simval=2,2\n--sigma_in=2,0\n--lambda=0.001,0.001\n--it=2,2\n--opt=DISCRETE,DISCRETE\n--CPgrid=1,2\n--SGgrid=3,4\n--datagrid=3,4\n--dopt=HOCR\n--VN\n--fixnan\n"
    with open(d + "conf", "w") as f:
        f.write(text)
    # irregular spheres: a subject's own sphere each, a template that is no regular icosphere (see test_group_multiresolution_matches_oracle)
    meshio.save_surface(d + "template.surf.gii", synthetic.known_warp(xyz, seed=33, rot_deg=7.0, amp=1.5), tri)
    subj = [synthetic.known_warp(xyz, seed=40 + s, rot_deg=0.0, amp=1.0) for s in range(S)]
    for s in range(S):
        meshio.save_surface(d + "sphere%d.surf.gii" % s, subj[s], tri)
    mask = (np.random.default_rng(1).random(len(xyz)) > 0.2).astype(np.float64)
    meshio.save_metric(d + "mask.func.gii", mask[None])
    datas = [synthetic.features(synthetic.known_warp(subj[s], seed=90 + s, rot_deg=3.0, amp=2.0), 2, seed=5) for s in range(S)]
    for s in range(S):
        meshio.save_metric(d + "data%d.func.gii" % s, datas[s])
    with open(d + "meshes.txt", "w") as f:
        f.write("".join(d + "sphere%d.surf.gii\n" % s for s in range(S)))
    with open(d + "data.txt", "w") as f:
        f.write("".join(d + "data%d.func.gii\n" % s for s in range(S)))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, "tools/register_files.py", "--groupwise", "--meshes=" + d + "meshes.txt", "--data=" + d + "data.txt",
                          "--template=" + d + "template.surf.gii", "--mask=" + d + "mask.func.gii", "--conf=" + d + "conf", "--out=" + d + "gw.", "--verbose"],
                         cwd=root, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr
    assert "Mesh #2 is" in run.stdout and "Template is" in run.stdout
    def sphere(path):
        p, _ = meshio.load_surface(path)
        p = p - p.mean(axis=0)
        return p * (100.0 / np.linalg.norm(p, axis=1, keepdims=True))

    in_xyz, t_xyz = [sphere(d + "sphere%d.surf.gii" % s) for s in range(S)], sphere(d + "template.surf.gii")
    fdatas = [meshio.load_data(d + "data%d.func.gii" % s, len(xyz)) for s in range(S)]
    cfg = config.parse_config(text)
    levels, run_kw, _ = config.levels_from_config(cfg, 2, groupwise=True)
    want, regs, _ = group_registration.run_group_multiresolution(group_registration.ProductGroupOps(ctx), [(p, tri) for p in in_xyz], fdatas, t_xyz, tri, levels,
                                                                 mask=meshio.load_data(d + "mask.func.gii", len(xyz))[0], fixnan=cfg["fixnan"], **run_kw)
    target = M.Mesh(ctx, t_xyz, tri)
    for s in range(S):
        reg, rtri = meshio.load_surface(d + "gw.sphere-%d.reg.surf.gii" % s)
        lr, lrtri = meshio.load_surface(d + "gw.sphere-%d.LR.reg.surf.gii" % s)
        assert np.array_equal(rtri, tri) and np.array_equal(lrtri, tri)
        assert np.abs(reg - want[s]).max() < 2e-5 and np.abs(lr - regs[-1][s]).max() < 2e-5 and angles(reg, in_xyz[s]).max() > 1e-4
        moved = meshio.load_data(d + "gw.transformed_and_reprojected-%d.func.gii" % s, len(xyz))
        direct = M.metric_resample(M.Mesh(ctx, want[s], tri), fdatas[s], target)
        assert moved.shape == (2, len(xyz)) and np.abs(moved - direct).max() < 1e-3 * max(1.0, np.abs(direct).max())
    # the run refuses what the reference refuses in this mode
    with open(d + "conf_affine", "w") as f:
        f.write(text.replace("--opt=DISCRETE,DISCRETE", "--opt=AFFINE,DISCRETE"))
    bad = subprocess.run([sys.executable, "tools/register_files.py", "--groupwise", "--meshes=" + d + "meshes.txt", "--data=" + d + "data.txt",
                          "--template=" + d + "template.surf.gii", "--conf=" + d + "conf_affine", "--out=" + d + "bad."], cwd=root, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "not supported in groupwise mode" in bad.stderr


def test_amsm_files_in_files_out(ctx, tmp_path):
    """tools/register_files.py with --inanat / --refanat and a --regoption=5 configuration (CLI/newmsm.cpp:40-47, set_anatomical M/mesh_registration.cpp:
    434-438): the outputs hold what run_multiresolution gives with the same anatomical surfaces; one anatomical mesh alone is refused."""
    import os
    import subprocess
    import sys

    from newmsm_amd import config, meshio

    xyz, tri = M.make_mesh_from_icosa(4)
    ref = synthetic.features(xyz, 1, 5)
    src = synthetic.features(synthetic.known_warp(xyz, seed=8, rot_deg=3.0, amp=2.0), 1, 5)
    ian, ran = synthetic.anatomy(xyz, seed=61, base=60.0), synthetic.anatomy(xyz, seed=71, base=62.0)
    d = str(tmp_path) + "/"
    text = ("--simval=2,2\n--sigma_in=2,0\n--sigma_ref=2,0\n--lambda=0.025,0.025\n--it=2,2\n--opt=DISCRETE,DISCRETE\n--CPgrid=1,2\n--SGgrid=3,4\n--datagrid=3,4\n"
            "--anatgrid=3,4\n--regoption=5\n--dopt=HOCR\n--triclique\n--rescaleL\n--shearmod=0.4\n--bulkmod=1.6\n--k_exponent=2\n")
    with open(d + "conf", "w") as f:
        f.write(text)
    meshio.save_surface(d + "sphere.surf.gii", xyz, tri)
    meshio.save_surface(d + "in.anat.surf.gii", ian, tri)
    meshio.save_surface(d + "ref.anat.surf.gii", ran, tri)
    meshio.save_metric(d + "in.func.gii", src)
    meshio.save_metric(d + "ref.func.gii", ref)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, "tools/register_files.py", "--inmesh=" + d + "sphere.surf.gii", "--indata=" + d + "in.func.gii", "--refdata=" + d + "ref.func.gii",
            "--conf=" + d + "conf", "--out=" + d + "a."]
    run = subprocess.run(base + ["--inanat=" + d + "in.anat.surf.gii", "--refanat=" + d + "ref.anat.surf.gii"], cwd=root, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr
    in_xyz, _ = meshio.load_surface(d + "sphere.surf.gii")
    in_xyz = in_xyz - in_xyz.mean(axis=0)
    in_xyz = in_xyz * (100.0 / np.linalg.norm(in_xyz, axis=1, keepdims=True))
    levels, run_kw, _ = config.levels_from_config(config.parse_config(text), 1, anat=True)
    want, _, _ = registration.run_multiresolution(registration.ProductOps(ctx), in_xyz, tri, meshio.load_data(d + "in.func.gii", len(xyz)), in_xyz, tri,
                                                  meshio.load_data(d + "ref.func.gii", len(xyz)), levels, in_anat=meshio.load_surface(d + "in.anat.surf.gii")[0],
                                                  ref_anat=meshio.load_surface(d + "ref.anat.surf.gii")[0], **run_kw)
    reg, _ = meshio.load_surface(d + "a.sphere.reg.surf.gii")
    assert np.abs(reg - want).max() < 2e-5 and angles(reg, xyz).max() > 1e-4
    plain, _, _ = registration.run_multiresolution(registration.ProductOps(ctx), in_xyz, tri, meshio.load_data(d + "in.func.gii", len(xyz)), in_xyz, tri,
                                                   meshio.load_data(d + "ref.func.gii", len(xyz)), [dict(lv, rmode=3) for lv in levels], **run_kw)
    assert np.abs(plain - want).max() > 1e-3  # the anatomical surfaces were used: the spherical regulariser ends elsewhere
    one = subprocess.run(base + ["--inanat=" + d + "in.anat.surf.gii"], cwd=root, capture_output=True, text=True, timeout=600)
    assert one.returncode != 0 and "must supply both anatomical meshes or none" in one.stderr
    none = subprocess.run(base, cwd=root, capture_output=True, text=True, timeout=600)
    assert none.returncode != 0 and "--regoption 5 requires anatomical meshes" in none.stderr


@pytest.mark.parametrize("kind,D", [("ho_univariate", 1), ("ho_multivariate", 16), ("univariate", 1)])
def test_fusion_driven_level_matches_oracle(ctx, kind, D):
    """The label loop of Fusion::optimize (I/Fusion/Fusion.h:136-229) over the triclique classes -- what every HCP configuration runs: per
    label step ONE fusion move (8 T triplet costs for the evolving labeling), here with a stand-in for the licence-restricted binary solve
    (iterated conditional modes, the same for both runs).  The MI355X path and the oracle take the same decisions in every one of the
    2 x L x iterations steps and end within the north star's 1e-4 rad.  ("univariate", 1) is BASELINE config 2's shape: the unary table +
    strain-only fusion moves (k_triplet_octets_packed).)"""
    xyz, tri, ref, src = level_inputs(4, D, seed=27)
    hcp = dict(lambda_=0.01, mu=0.4, kappa=1.6, k_exp=2.0, rexp=2.0)  # --shearmod --bulkmod --k_exponent --regexp of the HCP configurations
    kw = dict(cp_order=2, iters=2, seed=5, kind=kind, rescale_labels=True, cost_params=hcp, optimiser="fusion")
    t = {}
    got = registration.run_discrete_level(registration.ProductOps(ctx), xyz, tri, ref, xyz, tri, src, xyz, timings=t, **kw)
    want = registration.run_discrete_level(OracleOps(M.mcmc_optimise), xyz, tri, ref, xyz, tri, src, xyz, **kw)
    for a, b in zip(got[3], want[3]):
        assert np.array_equal(a, b)
    assert np.allclose(got[2], want[2], rtol=1e-9)
    assert "fusion_moves" in t and "triplet_table" not in t  # the fusion path, not the T x L^3 tables
    assert angles(got[0], xyz).max() > 1e-3 and any(l.any() for l in got[3])
    assert angles(got[0], want[0]).max() <= NORTH_STAR_TOL_RAD
    assert np.abs(got[0] - want[0]).max() < 1e-9


# ------------------------------------------------------------------------------------------------------------------------
# Full size: the ico6 subject and schedules bench.py times (BASELINE configs 2 and 3), two iterations per level, over the MI355X
# path and over the oracle -- the optimiser must take the same decisions in every iteration of every level, and the registered
# spheres must agree within north_star's 1e-4 rad.  (M/mesh_registration.cpp:30-50,164-232, I/Fusion/Fusion.h:136-229.)
# ------------------------------------------------------------------------------------------------------------------------
from helpers import registration_parity  # noqa: E402


@pytest.mark.parametrize("optimiser", ["mcmc", "fusion"])
def test_full_size_three_level_schedule_matches_oracle(ctx, optimiser):
    """bench.py's `registration` (Monte Carlo optimiser over the unary + T x L^3 tables) and `registration_fusion` (BASELINE config 2 as
    --dopt=HOCR drives it: unary table + strain-only fusion moves): data ico4/5/6, control ico2/3/4, sigma 4/2/1, --VN, sulc-like D = 1"""
    iters = (1, 1, 1) if optimiser == "mcmc" else (2, 2, 2)  # the oracle's T x L^3 triplet tables of the MCMC run are its slow part
    r = registration_parity(ctx, registration.basic_levels(iters), 1, mciters=50, mcparam=0.8, seed=1, cost_params=dict(lambda_=0.1), optimiser=optimiser)
    assert r["labelings"] == sum(iters) and r["labelings_identical"], r
    assert r["max_angle_rad"] <= NORTH_STAR_TOL_RAD and r["moved_rad"] > 1e-3, r
    assert r["energies_rel_diff"] < 1e-9, r


def test_full_size_msmall_schedule_matches_oracle(ctx):
    """BASELINE config 3: the HCP MSMAll schedule (triclique cost over 32 features, rescaled labels, --VN, fusion moves) at ico6,
    two iterations per level instead of 10 / 15 / 15"""
    r = registration_parity(ctx, registration.hcp_msmall_levels((2, 2, 2)), 32)
    assert r["labelings"] == 6 and r["labelings_identical"], r
    assert r["max_angle_rad"] <= NORTH_STAR_TOL_RAD and r["moved_rad"] > 1e-3, r
    assert r["energies_rel_diff"] < 1e-9, r


def test_pairwise_regoption1_level_matches_oracle(ctx):
    """--regoption=1 as --dopt=FastPD drives it (M/mesh_registration.cpp:182-188; BASELINE config 1's shape): per iteration the unary table and
    the P x L x L pair tables (computePairwiseCosts), then a stand-in for FPD::FastPD (iterated conditional modes, the same for both runs)"""
    xyz, tri, ref, src = level_inputs(4, 1, seed=23)
    kw = dict(cp_order=2, iters=3, seed=5, kind="univariate", rmode=1, cost_params=dict(lambda_=0.1, rexp=2.0), optimiser="fastpd")
    t = {}
    got = registration.run_discrete_level(registration.ProductOps(ctx), xyz, tri, ref, xyz, tri, src, xyz, timings=t, **kw)
    want = registration.run_discrete_level(OracleOps(M.mcmc_optimise), xyz, tri, ref, xyz, tri, src, xyz, **kw)
    assert len(got[3]) == 3 and "pairwise_table" in t and "triplet_table" not in t and "fusion_moves" not in t
    for a, b in zip(got[3], want[3]):
        assert np.array_equal(a, b)
    assert np.allclose(got[2], want[2], rtol=1e-9)
    assert angles(got[0], xyz).max() > 1e-3 and any(l.any() for l in got[3])
    assert angles(got[0], want[0]).max() <= NORTH_STAR_TOL_RAD and np.abs(got[0] - want[0]).max() < 1e-9
    with pytest.raises(ValueError, match="higher order clique regularisers with fastPD"):
        registration.run_discrete_level(registration.ProductOps(ctx), xyz, tri, ref, xyz, tri, src, xyz, cp_order=2, rmode=3, optimiser="fastpd")


def test_config_driven_registration_matches_oracle(ctx):
    """a configuration in the reference's grammar (the keys of config/basic_configs/config_standard_MSMpair, smaller grids) -> levels ->
    run_multiresolution over both paths; the AFFINE level is reported as skipped"""
    from newmsm_amd import config

    text = """
--sigma_in=4,4,2
--sigma_ref=4,4,2
--lambda=0,0.1,0.2
--it=50,2,2
--opt=AFFINE,DISCRETE,DISCRETE
--CPgrid=0,1,2
--SGgrid=0,3,4
--datagrid=3,3,4
--regoption=1
"""
    levels, run_kw, skipped = config.levels_from_config(config.parse_config(text), D=1)
    assert skipped == [(0, "AFFINE")] and len(levels) == 2
    in_xyz, in_tri = M.make_mesh_from_icosa(4)
    ref = synthetic.features(in_xyz, 1, 51)
    src = synthetic.features(synthetic.known_warp(in_xyz, seed=53, rot_deg=4.0, amp=2.5), 1, 51)
    lg, lw = [], []
    got = registration.run_multiresolution(registration.ProductOps(ctx), in_xyz, in_tri, src, in_xyz, in_tri, ref, levels, labelings_out=lg, **run_kw)
    want = registration.run_multiresolution(OracleOps(M.mcmc_optimise), in_xyz, in_tri, src, in_xyz, in_tri, ref, levels, labelings_out=lw, **run_kw)
    assert len(lg) == 4 and all(np.array_equal(a, b) for a, b in zip(lg, lw))
    assert angles(got[0], want[0]).max() <= NORTH_STAR_TOL_RAD and angles(got[0], in_xyz).max() > 1e-3


AMSM = dict(lambda_=0.025, mu=0.4, kappa=1.6, k_exp=2.0, rexp=2.0)  # config/NeuroImage2017_configs/aMSM_STR_longitudinal_alignment


def amsm_anatomy(ctx_or_none, xyz, tri, ops):
    """the anat argument of run_discrete_level for a synthetic subject: input / reference anatomical surfaces on the vertices of the data sphere"""
    return dict(in_anat=synthetic.anatomy(xyz, seed=61, base=60.0), ref_anat=synthetic.anatomy(xyz, seed=71, base=62.0), in_mesh=ops.mesh(xyz, tri), ref_mesh=ops.mesh(xyz, tri))


@pytest.mark.parametrize("kind,D,data_order,cp_order,anat_order,iters", [("ho_univariate", 1, 4, 2, 4, 2), ("univariate", 1, 4, 2, 3, 2), ("ho_univariate", 1, 6, 4, 6, 1)])
def test_amsm_level_matches_oracle(ctx, kind, D, data_order, cp_order, anat_order, iters):
    """--regoption=5 (aMSM) as a level runs it (VERDICT r3 missing 2): initialize_level's resample_anatomy (M/mesh_registration.cpp:91-99, 250-332: the control
    grid retessellated to --anatgrid with the face neighbourhoods -> NEARESTFACES, _ANATbaryweights in the reference's overwrite order, both anatomies
    resampled onto it by surface_resample) feeding the anatomical strain of every fusion move (computeTripletCost :169-182, deform_anatomy :255-301), driven
    as --dopt=HOCR drives it.  The MI355X path and the oracle take the same decisions in every label step and end within 1e-4 rad -- at ico4 / ico2 and for
    one full-size iteration (ico6 data, ico4 control grid, ico6 anatomical sphere: 40 960 evaluations x 16 faces per move)."""
    xyz, tri, ref, src = level_inputs(data_order, D, seed=33)
    pops, oops = registration.ProductOps(ctx), OracleOps(M.mcmc_optimise)
    kw = dict(cp_order=cp_order, iters=iters, seed=5, kind=kind, rescale_labels=True, cost_params=AMSM, optimiser="fusion", rmode=5)
    t = {}
    got = registration.run_discrete_level(pops, xyz, tri, ref, xyz, tri, src, xyz, timings=t, anat=dict(amsm_anatomy(ctx, xyz, tri, pops), order=anat_order), **kw)
    want = registration.run_discrete_level(oops, xyz, tri, ref, xyz, tri, src, xyz, anat=dict(amsm_anatomy(None, xyz, tri, oops), order=anat_order), **kw)
    assert len(got[3]) == iters
    for a, b in zip(got[3], want[3]):
        assert np.array_equal(a, b)
    assert np.allclose(got[2], want[2], rtol=1e-9)
    assert angles(got[0], want[0]).max() <= NORTH_STAR_TOL_RAD and np.abs(got[0] - want[0]).max() < 1e-9
    assert angles(got[0], xyz).max() > 1e-3 and t["surface_resample"] > 0
    # the regulariser is the anatomical one: the same level under the spherical strain (regoption 3) decides differently somewhere
    plain = registration.run_discrete_level(pops, xyz, tri, ref, xyz, tri, src, xyz, **dict(kw, rmode=3))
    assert not np.allclose(plain[2], got[2], rtol=1e-6)


def test_amsm_needs_the_anatomical_surfaces(ctx):
    xyz, tri, ref, src = level_inputs(4, 1, seed=33)
    with pytest.raises(ValueError, match="requires anatomical meshes"):
        registration.run_discrete_level(registration.ProductOps(ctx), xyz, tri, ref, xyz, tri, src, xyz, cp_order=2, iters=1, kind="univariate", cost_params=AMSM,
                                        optimiser="fusion", rmode=5)
    with pytest.raises(ValueError, match="both anatomical meshes or none"):
        registration.run_multiresolution(registration.ProductOps(ctx), xyz, tri, src, xyz, tri, ref, [dict(data_order=4, cp_order=2)], in_anat=synthetic.anatomy(xyz))


def test_amsm_preset_end_to_end_matches_oracle(ctx):
    """config/NeuroImage2017_configs/aMSM_STR_longitudinal_alignment (--regoption=5 --anatgrid=4,5,6 --triclique --dopt=HOCR) read through the grammar, with
    the anatomical surfaces given: two of its levels at one iteration each, the MI355X path against the oracle over run_multiresolution"""
    from newmsm_amd import config

    xyz, tri = M.make_mesh_from_icosa(5)
    ref = synthetic.features(xyz, 1, 7)
    src = synthetic.features(synthetic.known_warp(xyz, seed=9, rot_deg=3.0, amp=2.0), 1, 7)
    ian, ran = synthetic.anatomy(xyz, seed=61, base=60.0), synthetic.anatomy(xyz, seed=71, base=62.0)
    levels, run_kw, skipped = config.preset_levels("aMSM_STR", 1, iterations=(1, 1, 1), anat=True)
    levels = levels[:2]
    assert [lv["rmode"] for lv in levels] == [5, 5] and [lv["anat_order"] for lv in levels] == [4, 5] and skipped == []
    lg, lw = [], []
    got = registration.run_multiresolution(registration.ProductOps(ctx), xyz, tri, src, xyz, tri, ref, levels, labelings_out=lg, in_anat=ian, ref_anat=ran, **run_kw)
    want = registration.run_multiresolution(OracleOps(M.mcmc_optimise), xyz, tri, src, xyz, tri, ref, levels, labelings_out=lw, in_anat=ian, ref_anat=ran, **run_kw)
    assert len(lg) == 2 and all(np.array_equal(a, b) for a, b in zip(lg, lw))
    assert angles(got[0], want[0]).max() <= NORTH_STAR_TOL_RAD and angles(got[0], xyz).max() > 1e-3


def _cpp_newmsm():
    """tools/cpp/newmsm, built on demand (g++ over the headers of include/ and libmsmhip.so)"""
    import __graft_entry__ as g

    return g.build_cpp_newmsm()


def _same_files(a_prefix, b_prefix, names):
    for n in names:
        with open(a_prefix + n, "rb") as fa, open(b_prefix + n, "rb") as fb:
            assert fa.read() == fb.read(), "%s differs between the two programs" % n


@pytest.mark.parametrize("fmt", ["GIFTI", "ASCII", "ASCII_MAT"])
def test_cpp_newmsm_writes_the_same_files_as_the_python_tool(ctx, tmp_path, fmt):
    """tools/cpp/newmsm -- the `newmsm` executable with the host side in C++ (flags of src/msmOptions.h:59-157, CLI/newmsm.cpp:29-58) -- against
    tools/register_files.py on the same inputs: pairwise mode with cost-function weightings and the shipped standard_MSM_strain schedule (its AFFINE level
    skipped with a note), short and long flags, every output format: the three outputs byte for byte."""
    import os
    import subprocess
    import sys

    from newmsm_amd import config, meshio

    exe = _cpp_newmsm()
    xyz, tri = M.make_mesh_from_icosa(5)
    ref = synthetic.features(xyz, 2, 5)
    src = synthetic.features(synthetic.known_warp(xyz, seed=8, rot_deg=3.0, amp=2.0), 2, 5)
    d = str(tmp_path) + "/"
    text = config.PRESETS["standard_MSM_strain"].replace("--it=50,20,25,25", "--it=50,2,2,2").replace("--datagrid=5,5,5,6", "--datagrid=5,4,5,5").replace("--SGgrid=0,4,5,6", "--SGgrid=0,4,5,5").replace("--CPgrid=0,2,3,4", "--CPgrid=0,2,3,3")
    with open(d + "conf", "w") as f:
        f.write(text)
    meshio.save_surface(d + "in.surf.gii", synthetic.known_warp(xyz, seed=3, rot_deg=0.0, amp=0.7) + 0.25, tri)  # off-centre and irregular: recentre / rescale matter
    meshio.save_surface(d + "ref.surf.gii", xyz, tri)
    meshio.save_metric(d + "in.func.gii", src)
    meshio.save_metric(d + "ref.func.gii", ref)
    w = 0.5 + np.random.default_rng(2).random((1, len(xyz)))
    meshio.save_metric(d + "inw.func.gii", w)
    meshio.save_metric(d + "refw.func.gii", w[:, ::-1].copy())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--inmesh=" + d + "in.surf.gii", "--refmesh=" + d + "ref.surf.gii", "--indata=" + d + "in.func.gii", "--refdata=" + d + "ref.func.gii",
              "--inweight=" + d + "inw.func.gii", "--refweight=" + d + "refw.func.gii", "--conf=" + d + "conf", "-f", fmt]
    py = subprocess.run([sys.executable, "tools/register_files.py"] + common + ["--out=" + d + "py."], cwd=root, capture_output=True, text=True, timeout=600)
    assert py.returncode == 0, py.stderr
    cpp = subprocess.run([exe] + common + ["-o", d + "cpp.", "-v"], cwd=root, capture_output=True, text=True, timeout=600)
    assert cpp.returncode == 0, cpp.stderr
    assert "level 1 (--opt=AFFINE)" in cpp.stderr and "skipped" in cpp.stderr and "This is newMSM" in cpp.stdout
    surf, data = {"GIFTI": (".surf.gii", ".func.gii"), "ASCII": (".asc", ".dpv"), "ASCII_MAT": (".asc", ".txt")}[fmt]
    _same_files(d + "py.", d + "cpp.", ["sphere.reg" + surf, "sphere.LR.reg" + surf, "transformed_and_reprojected" + data])
    reg, _ = meshio.load_surface(d + "cpp.sphere.reg" + surf)
    assert angles(reg, meshio.load_surface(d + "in.surf.gii")[0]).max() > 1e-4  # something was registered
    # the error behaviour of CLI/newmsm.cpp:41-43,62-68: the message, exit status 1
    bad = subprocess.run([exe] + common + ["-o", d + "bad.", "--inanat=" + d + "in.surf.gii"], cwd=root, capture_output=True, text=True, timeout=600)
    assert bad.returncode == 1 and "must supply both anatomical meshes or none" in bad.stderr
    bad = subprocess.run([exe, "--nonsense"], cwd=root, capture_output=True, text=True, timeout=600)
    assert bad.returncode == 1 and "is not an option" in bad.stderr


def test_cpp_newmsm_groupwise_writes_the_same_files_as_the_python_tool(ctx, tmp_path):
    """the same for -g / --groupwise (CLI/newmsm.cpp:13-27): three subjects on spheres of their own, an irregular template, a mask, two levels: the three
    outputs per subject byte for byte; AFFINE levels refused with the reference's message"""
    import os
    import subprocess
    import sys

    from newmsm_amd import meshio

    exe = _cpp_newmsm()
    S = 3
    xyz, tri = M.make_mesh_from_icosa(4)
    d = str(tmp_path) + "/"
    text = "--simval=2,2\n--sigma_in=2,0\n--lambda=0.001,0.001\n--it=2,2\n--opt=DISCRETE,DISCRETE\n--CPgrid=1,2\n--SGgrid=3,4\n--datagrid=3,4\n--dopt=HOCR\n--VN\n--fixnan\n"
    with open(d + "conf", "w") as f:
        f.write(text)
    meshio.save_surface(d + "template.surf.gii", synthetic.known_warp(xyz, seed=33, rot_deg=7.0, amp=1.5), tri)
    subj = [synthetic.known_warp(xyz, seed=40 + s, rot_deg=0.0, amp=1.0) for s in range(S)]
    for s in range(S):
        meshio.save_surface(d + "sphere%d.surf.gii" % s, subj[s], tri)
        meshio.save_metric(d + "data%d.func.gii" % s, synthetic.features(synthetic.known_warp(subj[s], seed=90 + s, rot_deg=3.0, amp=2.0), 2, seed=5))
    meshio.save_metric(d + "mask.func.gii", (np.random.default_rng(1).random(len(xyz)) > 0.2).astype(np.float64)[None])
    with open(d + "meshes.txt", "w") as f:
        f.write("".join(d + "sphere%d.surf.gii\n" % s for s in range(S)))
    with open(d + "data.txt", "w") as f:
        f.write("".join(d + "data%d.func.gii\n" % s for s in range(S)))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--groupwise", "--meshes=" + d + "meshes.txt", "--data=" + d + "data.txt", "--template=" + d + "template.surf.gii", "--mask=" + d + "mask.func.gii", "--conf=" + d + "conf"]
    py = subprocess.run([sys.executable, "tools/register_files.py"] + common + ["--out=" + d + "py."], cwd=root, capture_output=True, text=True, timeout=600)
    assert py.returncode == 0, py.stderr
    cpp = subprocess.run([exe] + common + ["--out=" + d + "cpp."], cwd=root, capture_output=True, text=True, timeout=600)
    assert cpp.returncode == 0, cpp.stderr
    names = []
    for s in range(S):
        names += ["sphere-%d.reg.surf.gii" % s, "sphere-%d.LR.reg.surf.gii" % s, "transformed_and_reprojected-%d.func.gii" % s]
    _same_files(d + "py.", d + "cpp.", names)
    with open(d + "conf_affine", "w") as f:
        f.write(text.replace("--opt=DISCRETE,DISCRETE", "--opt=AFFINE,DISCRETE"))
    bad = subprocess.run([exe] + common[:-1] + ["--conf=" + d + "conf_affine", "--out=" + d + "bad."], cwd=root, capture_output=True, text=True, timeout=600)
    assert bad.returncode == 1 and "not supported in groupwise mode" in bad.stderr
