"""The C++ host side (include/msmhip.hpp, the reference's interface names over the C ABI) driven by a compiled C++
program (tests/cpp/host_mirror.cpp) with no Python in the loop; its results are compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from newmsm_amd import problem
from oracle import oracle as O
from tests.helpers import oracle_cost

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "host_mirror")
LIBDIR = os.path.join(ROOT, "newmsm_amd")


def build_cpp(src, exe):
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe, "-L", LIBDIR, "-lmsmhip",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)


def build_host_mirror():
    build_cpp(SRC, EXE)


LEVEL_SRC = os.path.join(ROOT, "tests", "cpp", "level_driver.cpp")
LEVEL_EXE = os.path.join(ROOT, "tests", "cpp", "level_driver")


from newmsm_amd.bag import read_bag, write_bag  # noqa: E402  (the container of the compiled programs' arrays)


GROUP_SRC = os.path.join(ROOT, "tests", "cpp", "group_driver.cpp")
GROUP_EXE = os.path.join(ROOT, "tests", "cpp", "group_driver")


def test_header_compiles_without_gpu(built):
    build_host_mirror()  # -Wall -Wextra -Werror: the header is clean C++17 and needs no HIP headers
    build_cpp(LEVEL_SRC, LEVEL_EXE)
    build_cpp(GROUP_SRC, GROUP_EXE)  # include/msmhip_group_registration.hpp


@pytest.mark.gpu
@pytest.mark.parametrize("D,rescale", [(1, 0), (3, 1)])
def test_cpp_level_driver_equals_python_loop(built, ctx, tmp_path, D, rescale):
    """run_discrete_opt of include/msmhip_registration.hpp (compiled, no Python) against newmsm_amd/registration.py: the same
    library calls in the same order with the same optimiser seeds, so labelings, energies and coordinates are identical"""
    import newmsm_amd as M
    from newmsm_amd import registration, synthetic

    build_cpp(LEVEL_SRC, LEVEL_EXE)
    xyz, tri = M.make_mesh_from_icosa(4)
    ref = synthetic.features(xyz, D, 21)
    src = synthetic.features(synthetic.known_warp(xyz, seed=23, rot_deg=4.0, amp=2.5), D, 21)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    write_bag(fin, orders=np.array([4, 2, D, 3, 40, 5, rescale]), params=np.array([0.3, 0.05]), ref_feat=ref, src_feat=src)
    run = subprocess.run([LEVEL_EXE, fin, fout], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr + run.stdout
    got = read_bag(fout)
    want = registration.run_discrete_level(registration.ProductOps(ctx), xyz, tri, ref, xyz, tri, src, xyz, cp_order=2, iters=3, mciters=40, mcparam=0.3,
                                           seed=5, kind="multivariate" if D > 1 else "univariate", rescale_labels=bool(rescale),
                                           cost_params=dict(lambda_=0.05))
    assert np.array_equal(got["labelings"].reshape(3, -1), np.array(want[3]))
    assert np.array_equal(got["energies"], np.array(want[2]))
    assert np.array_equal(got["sph_reg"].reshape(-1, 3), want[0]) and np.array_equal(got["cpgrid"].reshape(-1, 3), want[1])
    assert len({tuple(l) for l in want[3]}) > 1


@pytest.mark.gpu
@pytest.mark.parametrize("D", [1, 16])
def test_cpp_level_driver_fusion_loop_equals_python_loop(built, ctx, tmp_path, D):
    """the same for the fusion-driven loop over the triclique classes (LevelOptions::fusion: per label step one msm_cost_triplet_octets into
    a pinned buffer + the stand-in solve): identical to registration.run_discrete_level(optimiser="fusion")"""
    import newmsm_amd as M
    from newmsm_amd import registration, synthetic

    build_cpp(LEVEL_SRC, LEVEL_EXE)
    xyz, tri = M.make_mesh_from_icosa(4)
    ref = synthetic.features(xyz, D, 21)
    src = synthetic.features(synthetic.known_warp(xyz, seed=23, rot_deg=4.0, amp=2.5), D, 21)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    write_bag(fin, orders=np.array([4, 2, D, 2, 0, 5, 1, 1]), params=np.array([0.3, 0.01]), ref_feat=ref, src_feat=src)
    run = subprocess.run([LEVEL_EXE, fin, fout], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr + run.stdout
    got = read_bag(fout)
    want = registration.run_discrete_level(registration.ProductOps(ctx), xyz, tri, ref, xyz, tri, src, xyz, cp_order=2, iters=2, seed=5,
                                           kind="ho_multivariate" if D > 1 else "ho_univariate", rescale_labels=True, optimiser="fusion",
                                           cost_params=dict(lambda_=0.01, mu=0.4, kappa=1.6, k_exp=2.0, rexp=2.0))
    assert np.array_equal(got["labelings"].reshape(2, -1), np.array(want[3]))
    assert np.array_equal(got["energies"], np.array(want[2]))
    assert np.array_equal(got["sph_reg"].reshape(-1, 3), want[0]) and np.array_equal(got["cpgrid"].reshape(-1, 3), want[1])
    assert any(l.any() for l in want[3])


@pytest.mark.gpu
def test_cpp_level_driver_pairwise_loop_equals_python_loop(built, ctx, tmp_path):
    """LevelOptions::pairwise (--regoption=1 as --dopt=FastPD drives it: computeUnaryCosts, computePairwiseCosts, the stand-in solve): identical
    to registration.run_discrete_level(optimiser="fastpd")"""
    import newmsm_amd as M
    from newmsm_amd import registration, synthetic

    build_cpp(LEVEL_SRC, LEVEL_EXE)
    xyz, tri = M.make_mesh_from_icosa(4)
    ref = synthetic.features(xyz, 1, 21)
    src = synthetic.features(synthetic.known_warp(xyz, seed=23, rot_deg=4.0, amp=2.5), 1, 21)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    write_bag(fin, orders=np.array([4, 2, 1, 3, 0, 5, 0, 0, 1]), params=np.array([0.3, 0.1]), ref_feat=ref, src_feat=src)
    run = subprocess.run([LEVEL_EXE, fin, fout], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr + run.stdout
    got = read_bag(fout)
    want = registration.run_discrete_level(registration.ProductOps(ctx), xyz, tri, ref, xyz, tri, src, xyz, cp_order=2, iters=3, seed=5, kind="univariate",
                                           rmode=1, optimiser="fastpd", cost_params=dict(lambda_=0.1))
    assert np.array_equal(got["labelings"].reshape(3, -1), np.array(want[3]))
    assert np.array_equal(got["energies"], np.array(want[2]))
    assert np.array_equal(got["sph_reg"].reshape(-1, 3), want[0]) and np.array_equal(got["cpgrid"].reshape(-1, 3), want[1])
    assert any(l.any() for l in want[3])


@pytest.mark.gpu
@pytest.mark.parametrize("D", [1, 3])
def test_cpp_host_mirror_against_oracle(built, tmp_path, D):
    build_host_mirror()
    inp = problem.pairwise_inputs(4, 2, D=D)
    rng = np.random.default_rng(12)
    T, L, N = len(inp["triplets"]), len(inp["labels"]), len(inp["cp_xyz"])
    tq = [rng.integers(0, T, 200), rng.integers(0, L, 200), rng.integers(0, L, 200), rng.integers(0, L, 200)]
    labeling = rng.integers(0, L, N)
    folded_cp = np.array(inp["cp_orig_xyz"])
    for v, n in ((10, 11), (40, 41), (90, 12)):  # three control points pushed across a neighbour
        p = folded_cp[n] + 1.4 * (folded_cp[n] - folded_cp[v])
        folded_cp[v] = p * 100.0 / np.linalg.norm(p)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    write_bag(fin, orders=np.array([4, 2, D]), ref_feat=inp["ref_feat"], src_feat=inp["src_feat"], source_xyz=inp["source_xyz"],
              labels=inp["labels"], samples0=inp["samples"][0], folded_cp=folded_cp, tq_t=tq[0], tq_a=tq[1], tq_b=tq[2], tq_c=tq[3], labeling=labeling)
    run = subprocess.run([EXE, fin, fout], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr + run.stdout
    assert "expected error: Unknown similarity metric" in run.stdout
    got = read_bag(fout)
    assert got["error_code"][0] == -1

    # the oracle on the same iteration: the control grid is carried through the warp by sphere_project_warp first
    regular = O.Mesh(inp["source_orig_xyz"], inp["source_tri"])
    cp_now = O.sphere_project_warp(inp["cp_orig_xyz"], regular, inp["source_xyz"])
    assert np.array_equal(got["cp_now"].reshape(-1, 3), cp_now)
    cpm = O.Mesh(cp_now, inp["cp_tri"])
    maxsep, mvd = O.cp_spacings(cpm)
    inp2 = dict(inp, cp_xyz=cp_now, maxsep=maxsep, mvdmax=mvd, rot=O.cp_rotations(inp["samples"][0], cp_now))
    kind = "multivariate" if D > 1 else "univariate"
    oc = oracle_cost(inp2, kind, lambda_=0.2)
    oc.set_pairs(np.zeros((0, 2), dtype=np.int32))  # regoption 3: the model has triplets only
    oc.get_source_data()
    assert np.array_equal(got["absw"], oc.absolute_weights())
    U = oc.unary_table()
    assert np.allclose(got["unarycosts"].reshape(U.shape), U, rtol=1e-9, atol=1e-11)
    assert abs(got["single"][0] - U[3, 5]) <= 1e-11 + 1e-9 * abs(U[3, 5])
    want = np.array([oc.triplet(*q) for q in zip(*tq)])
    assert np.allclose(got["triplet"], want, rtol=1e-9, atol=1e-11)
    assert abs(got["single"][1] - oc.triplet(7, 1, 2, 3)) <= 1e-11 + 1e-9 * abs(got["single"][1])
    E = got["octets"].reshape(T, 8)
    for t in (0, 11, T - 1):
        ids = inp["triplets"][t]
        for k in range(8):
            la, lb, lc = (4 if k & 4 else labeling[ids[0]]), (4 if k & 2 else labeling[ids[1]]), (4 if k & 1 else labeling[ids[2]])
            assert abs(E[t, k] - oc.triplet(t, int(la), int(lb), int(lc))) <= 1e-11 + 1e-9 * abs(E[t, k])
    assert abs(got["total"][0] - oc.total(labeling.astype(np.int32))[0]) <= 1e-9 * abs(got["total"][0])
    src = O.Mesh(inp["source_xyz"], inp["source_tri"])
    assert np.array_equal(got["resampled"].reshape(D, -1), O.metric_resample(src, inp["src_feat"], regular))
    fm = O.Mesh(folded_cp, inp["cp_tri"])
    assert tuple(got["unfold_counts"]) == O.unfold(fm) and got["unfold_counts"][1] >= 1
    assert np.array_equal(got["unfolded_cp"].reshape(-1, 3), fm.xyz)
    assert np.array_equal(got["normed"].reshape(D, -1), O.variance_normalise(inp["src_feat"]))


# ---------------------------------------------------------------------------------------------------------------------
# The unmodified optimiser's call pattern against msmhip::FusionModel / GroupFusionModel (tests/cpp/fusion_replay.cpp)
FUSION_SRC = os.path.join(ROOT, "tests", "cpp", "fusion_replay.cpp")
FUSION_EXE = os.path.join(ROOT, "tests", "cpp", "fusion_replay")


def build_fusion_replay():
    cmd = ["g++", "-std=c++17", "-O1", "-fopenmp", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), FUSION_SRC, "-o", FUSION_EXE, "-L", LIBDIR,
           "-lmsmhip", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_fusion_replay_compiles_without_gpu(built):
    build_fusion_replay()


def pseudo_optimiser(labeling, label, rng_state):
    """the stand-in for ELC + FastPD of fusion_replay.cpp: a fixed pseudo-random third of the nodes takes the proposed label"""
    lab = labeling.copy()
    for node in range(len(lab)):
        rng_state = (rng_state * 1664525 + 1013904223) & 0xFFFFFFFF
        if lab[node] != label and (rng_state >> 24) % 3 == 0:
            lab[node] = label
    return lab, rng_state


@pytest.mark.gpu
@pytest.mark.parametrize("kind,D,rmode", [("ho_univariate", 1, 3), ("ho_multivariate", 16, 3), ("univariate", 1, 3), ("univariate", 1, 1)])
def test_fusion_loops_through_the_adapter(built, tmp_path, kind, D, rmode):
    """Fusion::optimize's three OpenMP loops (8 threads, 2 sweeps over all labels), unmodified, over msmhip::FusionModel: every buffer
    they fill equals the oracle's replay of the same calls, with ONE ABI call per label step and no clique evaluated on its own"""
    from newmsm_amd.api import KINDS

    build_fusion_replay()
    inp = problem.pairwise_inputs(4, 2, D=D)
    par = dict(lambda_=0.05, mu=0.4, kappa=1.6, k_exp=2.0, rexp=2.0)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    write_bag(fin, orders=np.array([0, 4, 2, D, 8, 2, KINDS[kind], rmode]), params=np.array([par["lambda_"], par["mu"], par["kappa"], par["k_exp"], par["rexp"]]),
              target_xyz=inp["target_xyz"], ref_feat=inp["ref_feat"], src_feat=inp["src_feat"], source_xyz=inp["source_xyz"], cp_xyz=inp["cp_xyz"],
              maxsep=inp["maxsep"], mvdmax=np.array([inp["mvdmax"]]), labels=inp["labels"], rot=inp["rot"], triplets=inp["triplets"], pairs=inp["pairs"])
    run = subprocess.run([FUSION_EXE, fin, fout], capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="8"))
    assert run.returncode == 0, run.stderr + run.stdout
    got = read_bag(fout)
    oc = oracle_cost(inp, kind, rmode=rmode, **par)
    if rmode == 1:
        oc.set_triplets(np.zeros((0, 3), dtype=np.int32))
    else:
        oc.set_pairs(np.zeros((0, 2), dtype=np.int32))
    oc.get_source_data()
    N, L, T, P = len(inp["cp_xyz"]), len(inp["labels"]), (len(inp["triplets"]) if rmode != 1 else 0), (len(inp["pairs"]) if rmode == 1 else 0)
    steps = got["steps"]
    labelings = got["labelings"].reshape(len(steps), N)
    assert len(steps) >= 2 * L - 2  # a step is skipped only when every node already holds its label
    U = oc.unary_table()
    lab, state = np.zeros(N, dtype=np.int32), 12345
    unary, trip, pairs = got["unary"].reshape(len(steps), N, 2), got["triplets"].reshape(len(steps), T, 8), got["pairs"].reshape(len(steps), P, 4)
    si = 0
    for sweep in range(2):
        for label in range(L):
            if np.abs(label - lab).sum() == 0:
                continue
            assert steps[si] == label and np.array_equal(labelings[si], lab)
            assert np.allclose(unary[si, :, 0], U[lab, np.arange(N)], rtol=1e-9, atol=1e-11) and np.allclose(unary[si, :, 1], U[label], rtol=1e-9, atol=1e-11)
            if T and si % 7 == 0:  # the oracle's replay of the 8 T calls (every seventh step: the oracle is the slow side)
                want = oc.triplet_octets(lab, label, threads=8)
                assert np.allclose(trip[si], want, rtol=1e-9, atol=1e-11), np.abs(trip[si] - want).max()
            if P and si % 7 == 0:
                pr = inp["pairs"]
                for p in range(0, P, 5):
                    a, b = int(lab[pr[p, 0]]), int(lab[pr[p, 1]])
                    want = [oc.pairwise(p, a, b), oc.pairwise(p, a, label), oc.pairwise(p, label, b), oc.pairwise(p, label, label)]
                    assert np.allclose(pairs[si, p], want, rtol=1e-9, atol=1e-11)
            lab, state = pseudo_optimiser(lab, label, state)
            si += 1
    assert si == len(steps)
    want_total = oc.total(lab)[0]
    assert abs(got["total"][0] - want_total) <= 1e-9 * abs(want_total) + 1e-11
    step_calls, single_calls, served = got["counts"]
    assert single_calls == 0
    assert step_calls == (len(steps) if T else 0)  # exactly one msm_cost_triplet_octets per label step; pair costs come from the table
    # msmhip::fusion_optimize (include/msmhip_fusion.hpp) over the same model with a stand-in PBF / solver: from whole-step buffers and
    # through the per-clique evaluators the same labelings; one ABI call per step taken; the energy it returns is the oracle's
    same, nsteps, abi_calls, energy, moved, singles = got["fused_info"]
    assert same == 1.0 and abi_calls == nsteps >= 2 * L - 2 and moved > 0 and singles == 0
    want_total = oc.total(got["fused_labeling"])[0]
    assert abs(energy - want_total) <= 1e-9 * abs(want_total) + 1e-11


@pytest.mark.gpu
def test_group_fusion_loops_through_the_adapter(built, tmp_path):
    """the same for gMSM: 4 P pair + 8 T triplet calls per label step from 8 threads over msmhip::GroupFusionModel -> one
    msm_group_fusion_move per step; buffers against the oracle's evaluators"""
    import newmsm_amd as M
    from newmsm_amd import synthetic

    build_fusion_replay()
    S, D, data_order, cp_order = 3, 2, 4, 2
    dxyz, dtri = M.make_mesh_from_icosa(data_order)
    cxyz, ctri = M.make_mesh_from_icosa(cp_order)
    _, mvd = M.cp_spacings(cxyz, ctri)
    samples, _ = M.label_sampling_grid(cp_order + 2, 0.5 * mvd)
    sph = np.stack([synthetic.known_warp(dxyz, seed=40 + s, rot_deg=1.0 + s, amp=0.5) for s in range(S)])
    feat = np.stack([synthetic.features(synthetic.known_warp(dxyz, seed=90 + s, rot_deg=2.0, amp=1.0), D, seed=5) for s in range(S)])
    cps = np.stack([synthetic.known_warp(cxyz, seed=40 + s, rot_deg=1.0 + s, amp=0.5) for s in range(S)])
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    write_bag(fin, orders=np.array([1, data_order, cp_order, D, 8, 1, S]), params=np.array([0.2]), labels=samples, sph=sph, feat=feat, cp=cps)
    run = subprocess.run([FUSION_EXE, fin, fout], capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="8"))
    assert run.returncode == 0, run.stderr + run.stdout
    got = read_bag(fout)
    og = O.Group(S, simmeasure=2, lambda_=0.2)
    keep = [O.Mesh(dxyz, dtri)]
    og.set_template(keep[0], None)
    og.set_controlgrid(O.Mesh(cxyz, ctri))
    for s in range(S):
        om = O.Mesh(dxyz, dtri)
        og.set_subject(s, om, feat[s])
        om.set_coords(sph[s])
        og.set_subject(s, om, feat[s])
        og.reset_cpgrid(s, cps[s])
        keep.append(om)
    og.set_labels(samples)
    og.setup()
    N, L, P, T = og.num_nodes, len(samples), og.P, og.T
    steps = got["steps"]
    labelings = got["labelings"].reshape(len(steps), N)
    quads, octs = got["pairs"].reshape(len(steps), P, 4), got["triplets"].reshape(len(steps), T, 8)
    pr, tr = og.pairs(), og.triplets()
    rng = np.random.default_rng(3)
    for si in range(0, len(steps), 5):
        lab, label = labelings[si], int(steps[si])
        for p in rng.integers(0, P, 40):
            a, b = int(lab[pr[p, 0]]), int(lab[pr[p, 1]])
            want = np.array([og.pairwise(int(p), a, b), og.pairwise(int(p), a, label), og.pairwise(int(p), label, b), og.pairwise(int(p), label, label)])
            assert np.allclose(quads[si, p], want, rtol=1e-9, atol=1e-11, equal_nan=True)
        for t in rng.integers(0, T, 40):
            for k in range(8):
                l3 = [label if k >> (2 - j) & 1 else int(lab[tr[t, j]]) for j in range(3)]
                w = og.triplet(int(t), *l3)
                assert abs(octs[si, t, k] - w) <= 1e-11 + 1e-9 * abs(w)
    step_calls, single_calls, served = got["counts"]
    assert single_calls == 0 and step_calls == len(steps)
    same, nsteps, abi_calls, energy, moved, singles = got["fused_info"]  # msmhip::fusion_optimize over the group model, see above
    assert same == 1.0 and abi_calls == nsteps >= 2 * L - 2 and singles == 0
    lab = got["fused_labeling"]
    want = sum(og.pairwise(p, int(lab[pr[p, 0]]), int(lab[pr[p, 1]])) for p in range(P)) + sum(og.triplet(t, *(int(lab[v]) for v in tr[t])) for t in range(T))
    assert (np.isnan(want) and np.isnan(energy)) or abs(energy - want) <= 1e-9 * abs(want) + 1e-11


MULTIRES_CONFIGS = {
    # two DISCRETE levels driven as --dopt=HOCR drives them; smoothing and --VN on (the featurespace calls of a level), the second level starts from the first one's warp
    "fusion": "--opt=DISCRETE,DISCRETE\n--simval=2,2\n--sigma_in=2,1\n--sigma_ref=2,1\n--lambda=0.1,0.1\n--it=2,2\n--CPgrid=1,2\n--SGgrid=3,4\n--datagrid=3,4\n--dopt=HOCR\n--regoption=3\n--VN\n",
    "triclique": "--opt=DISCRETE,DISCRETE\n--simval=2,2\n--sigma_in=0,0\n--sigma_ref=0,0\n--lambda=0.01,0.02\n--it=2,2\n--CPgrid=1,2\n--SGgrid=3,4\n--datagrid=3,4\n--dopt=HOCR\n--regoption=3\n"
                 "--triclique\n--rescaleL\n--shearmod=0.4\n--bulkmod=1.6\n--k_exponent=2\n--regexp=2\n",
    # --regoption=5 (aMSM): resample_anatomy per level (M/mesh_registration.cpp:250-332), the anatomical strain in every fusion move
    "amsm": "--opt=DISCRETE,DISCRETE\n--simval=2,2\n--sigma_in=2,1\n--sigma_ref=2,1\n--lambda=0.025,0.025\n--it=2,1\n--CPgrid=1,2\n--SGgrid=3,4\n--datagrid=3,4\n--anatgrid=3,4\n"
            "--dopt=HOCR\n--regoption=5\n--regexp=2\n--VN\n--rescaleL\n--triclique\n--k_exponent=2\n--bulkmod=1.6\n--shearmod=0.4\n",
}


@pytest.mark.gpu
@pytest.mark.parametrize("name,D", [("fusion", 1), ("triclique", 4), ("amsm", 1)])
def test_cpp_run_multiresolutions_equals_python_loop(built, ctx, tmp_path, name, D):
    """tools/cpp/registration_bench -- msmhip::run_multiresolutions with its schedule from msmhip_config.hpp, what bench.py's registration_*_cpp
    objects run -- against newmsm_amd/registration.py: run_multiresolution over config.py's reading of the same text: identical labelings in
    every iteration of every level, the same registered sphere (M/mesh_registration.cpp:30-50,131-232)."""
    import __graft_entry__ as g
    import newmsm_amd as M
    from newmsm_amd import config, registration, synthetic

    exe = g.build_cpp_host()
    xyz, tri = M.make_mesh_from_icosa(4)
    ref = synthetic.features(xyz, D, 7)
    src = synthetic.features(synthetic.known_warp(xyz, seed=9, rot_deg=3.0, amp=2.0), D, 7)
    fin, fout, conf = str(tmp_path / "in.bag"), str(tmp_path / "out.bag"), str(tmp_path / "conf")
    anat = {}
    if name == "amsm":
        anat = dict(in_anat=synthetic.anatomy(xyz, seed=61, base=60.0), ref_anat=synthetic.anatomy(xyz, seed=71, base=62.0))
    write_bag(fin, orders=np.array([4, D], dtype=np.int32), in_data=src, ref_data=ref, **anat)
    with open(conf, "w") as f:
        f.write(MULTIRES_CONFIGS[name])
    run = subprocess.run([exe, fin, fout, conf, "1"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr + run.stdout
    import json

    line = json.loads(run.stdout.strip().splitlines()[-1])
    got = read_bag(fout)
    levels, run_kw, skipped = config.levels_from_config(config.parse_config(MULTIRES_CONFIGS[name]), D, anat=bool(anat))
    labs = []
    sphere, regs, energies = registration.run_multiresolution(registration.ProductOps(ctx), xyz, tri, src, xyz, tri, ref, levels, labelings_out=labs, **run_kw, **anat)
    assert line["levels"] == 2 and len(labs) == (3 if name == "amsm" else 4) and list(got["nodes"]) == [len(l) for l in labs]
    assert np.array_equal(got["labelings"], np.concatenate(labs))
    assert np.allclose(got["energies"], np.concatenate(energies), rtol=1e-12, atol=0)
    assert np.allclose(got["sphere_reg"].reshape(-1, 3), sphere, rtol=0, atol=1e-10)
    assert any(np.any(l != 0) for l in labs)   # the registration moved something
    assert line["moves"] == line["calls"]["fusion_moves"] > 0 and line["moves_timed"] == line["moves"] and line["move_kernel_us"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("levels_from", ["arrays", "config"])
def test_cpp_group_multiresolutions_equals_python_loop(built, ctx, tmp_path, levels_from):
    """run_group_multiresolutions of include/msmhip_group_registration.hpp (compiled, no Python: tests/cpp/group_driver.cpp) against
    newmsm_amd/group_registration.py: run_group_multiresolution -- Group_Mesh_registration's level loop (M/group_mesh_registration.cpp:26-133) with the
    host side in C++: two levels, three subjects on irregular spheres, a --mask, variance normalisation; the same library calls in the same order, so
    the labelings of all four iterations are identical and the registered spheres agree to rounding of the energy sums (which decide nothing here).
    "config": the levels of both runs come from a configuration text -- group_levels_from_config in C++, config.levels_from_config(groupwise=True) here."""
    import newmsm_amd as M
    from newmsm_amd import group_registration as GR, synthetic

    build_cpp(GROUP_SRC, GROUP_EXE)
    S, D = 3, 2
    xyz, tri = M.make_mesh_from_icosa(4)
    txyz = synthetic.known_warp(xyz, seed=33, rot_deg=7.0, amp=1.5)
    meshes = [(synthetic.known_warp(xyz, seed=40 + s, rot_deg=0.0, amp=1.0), tri) for s in range(S)]
    datas = [synthetic.features(synthetic.known_warp(meshes[s][0], seed=90 + s, rot_deg=3.0, amp=2.0), D, seed=5) for s in range(S)]
    levels = [dict(data_order=3, cp_order=1, sg_order=3, iters=2, simmeasure=2, cost_params=dict(lambda_=1e-3), sigma_in=2.0),
              dict(data_order=4, cp_order=2, sg_order=4, iters=2, simmeasure=2, cost_params=dict(lambda_=1e-3), sigma_in=0.0)]
    mask = (np.random.default_rng(1).random(len(xyz)) > 0.2).astype(np.float64)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    arrays = dict(sizes=np.array([S, D, len(levels), 1, 1, 1], dtype=np.int32), template_xyz=txyz, template_tri=tri.astype(np.int32), mask=mask,
                  level_orders=np.array([[lv["data_order"], lv["cp_order"], lv["sg_order"], lv["iters"], lv["simmeasure"]] for lv in levels], dtype=np.int32),
                  level_params=np.array([[lv["sigma_in"], lv["cost_params"]["lambda_"]] for lv in levels]))
    for s in range(S):
        arrays.update({"mesh%d_xyz" % s: meshes[s][0], "mesh%d_tri" % s: tri.astype(np.int32), "data%d" % s: datas[s]})
    write_bag(fin, **arrays)
    cmd, kw = [GROUP_EXE, fin, fout], dict(varnorm=True, fixnan=True)
    if levels_from == "config":
        from newmsm_amd import config

        text = ("--simval=2,2\n--sigma_in=2,0\n--lambda=0.001,0.001\n--it=2,2\n--opt=DISCRETE,DISCRETE\n--CPgrid=1,2\n--SGgrid=3,4\n--datagrid=3,4\n--dopt=HOCR\n--VN\n--fixnan\n"
                "--shearmod=0.4\n--bulkmod=1.6\n")
        conf = str(tmp_path / "conf")
        with open(conf, "w") as f:
            f.write(text)
        cmd.append(conf)
        cfg = config.parse_config(text)
        levels, run_kw, _ = config.levels_from_config(cfg, D, groupwise=True)
        kw = dict(fixnan=cfg["fixnan"], **run_kw)
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr + run.stdout
    got = read_bag(fout)
    labs = []
    want = GR.run_group_multiresolution(GR.ProductGroupOps(ctx), meshes, datas, txyz, tri, levels, mask=mask, labelings_out=labs, **kw)
    assert len(labs) == 4 and np.array_equal(got["labelings"], np.concatenate(labs)) and any(l.any() for l in labs[2:])
    assert np.allclose(got["energies"], np.concatenate(want[2]), rtol=1e-12, atol=0)
    for s in range(S):
        assert np.allclose(got["sphere_reg%d" % s].reshape(-1, 3), want[0][s], rtol=0, atol=1e-10)
        assert np.allclose(got["level_reg%d" % s].reshape(-1, 3), want[1][-1][s], rtol=0, atol=1e-10)
        assert np.abs(want[0][s] - meshes[s][0]).max() > 1e-3
