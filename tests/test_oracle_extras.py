"""Known-answer checks of the oracle's later additions (CPU only): the DICE measures, smooth_data and the
anatomical strain regulariser.  The reference ships no vectors for them; these pin the
restatement to properties that follow from the reference's formulas."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import oracle_anatomy


def test_dice_known_answers():
    rng = np.random.default_rng(0)
    a = rng.normal(size=40)
    # identical vectors: the same 25 % of the elements is above both thresholds -> 1 - 2c/(c + c) = 0
    assert O.sim_for_min(4, a, a, None, 0.75) == 0.0
    assert O.sim_for_min(5, a, a, None, 0.75) == 0.0
    # reversed order statistics: the top quarters are disjoint -> 1
    b = -a
    assert O.sim_for_min(4, a, b, None, 0.75) == 1.0
    # hand count, M/similarities.cpp:201-226 with idx = floor(0.5 * 6) = 3
    A = np.array([5.0, 1.0, 4.0, 2.0, 6.0, 3.0])   # sorted 1 2 3 4 5 6 -> threshold 4 -> {5,4,6} at ids 0,2,4
    B = np.array([9.0, 8.0, 1.0, 2.0, 7.0, 3.0])   # sorted 1 2 3 7 8 9 -> threshold 7 -> ids 0,1,4
    # size_A = 3, size_B = 3, common = ids {0,4} = 2 -> 1 - 4/6
    assert O.sim_for_min(4, A, B, None, 0.5) == 1.0 - (2.0 * 2) / 6
    assert abs(O.sim_for_min(5, A, B, None, 0.5) - (1.0 - 2.0 * ((2 / 9.0) / (6 / 9.0)))) < 1e-16


def test_smooth_data_properties():
    xyz, tri = O.icosphere(3)
    m = O.Mesh(xyz, tri)
    const = np.full((1, len(xyz)), 3.25)
    out = O.smooth_data(m, const, m, 12.0)
    assert np.allclose(out, 3.25, rtol=1e-14, atol=0)          # a weighted mean of a constant
    # the icosphere's neighbourhoods are point-symmetric enough that a linear function stays (nearly) proportional
    lin = xyz[:, :1].T.copy()
    sm = O.smooth_data(m, lin, m, 12.0)[0]
    k = (sm @ lin[0]) / (lin[0] @ lin[0])
    assert 0.9 < k < 1.0 and np.max(np.abs(sm - k * lin[0])) < 5.0  # 5- and 6-valent vertices differ a little
    # sigma -> tiny: only the vertex itself is in range -> identity
    rng = np.random.default_rng(1)
    noise = rng.normal(size=(2, len(xyz)))
    assert np.allclose(O.smooth_data(m, noise, m, 0.05), noise, rtol=4e-16, atol=0)  # (x * w) / w
    # exclusion: excluded centres are zeroed, excluded neighbours carry no weight, the mask output is the kept weight share
    excl = np.ones(len(xyz))
    excl[::5] = 0.0
    out, mask = O.smooth_data(m, const, m, 12.0, excl)
    assert np.all(out[0, excl == 0] == 0.0) and np.allclose(out[0, excl > 0], 3.25, rtol=1e-14)
    assert np.all(mask[excl == 0] == 0.0) and np.all((mask[excl > 0] > 0.3) & (mask[excl > 0] <= 1.0)) and mask[excl > 0].mean() < 0.9


def test_anatomical_strain_known_answers():
    cxyz, ctri, axyz, atri, w_ptr, w_cp, w_val, face_ptr, face_idx = oracle_anatomy()
    cp = O.Mesh(cxyz, ctri)
    data = O.Mesh(*O.icosphere(3))
    c = O.Cost("univariate", rmode=5, lambda_=1.0, mu=0.4, kappa=1.6, rexp=1.0)
    c.set_meshes(data, O.Octree(data), data, cp)
    c.set_features(np.zeros((1, data.V)), np.zeros((1, data.V)))
    labels = np.array([[0.0, 0.0, 100.0], [1.5, 0.0, np.sqrt(1e4 - 2.25)]])
    rot = O.cp_rotations(labels[0], cxyz)
    c.set_labels(labels, rot)
    c.set_triplets(O.estimate_triplets(cp))
    sphere = O.Mesh(axyz, atri)
    radial = 60.0 + 5.0 * np.cos(axyz[:, 2] / 40.0)
    anat = axyz / 100.0 * radial[:, None]
    asrc = O.Mesh(anat, atri)
    # target anatomy == source anatomy and no displacement (label 0 everywhere): every anatomical vertex comes back to itself
    c.set_anatomical(sphere, O.Octree(sphere), anat, asrc, w_ptr, w_cp, w_val, face_ptr, face_idx)
    for t in (0, 7, 100):
        assert abs(c.triplet(t, 0, 0, 0)) < 1e-10
    # a uniformly scaled target anatomy: every face grows by s in both directions -> the same strain density everywhere,
    # so all triplets agree and the cost grows with the scale factor
    costs = []
    for s in (1.05, 1.2):
        c.set_anatomical(sphere, O.Octree(sphere), anat * s, asrc, w_ptr, w_cp, w_val, face_ptr, face_idx)
        v = [c.triplet(t, 0, 0, 0) for t in (0, 7, 100)]
        assert np.ptp(v) < 1e-9 * max(v) and v[0] > 0
        costs.append(v[0])
    assert costs[1] > costs[0]
    # moving a control point changes only the triplets that contain it
    t_with = int(np.nonzero((O.estimate_triplets(cp) == 5).any(axis=1))[0][0])
    assert c.triplet(t_with, 1, 0, 0) != c.triplet(t_with, 0, 0, 0)
