"""Mesh_registration::resample_anatomy (M/mesh_registration.cpp:250-332: what a --regoption=5 level prepares for the anatomical regulariser) on the
host: the library's msm_resample_anatomy_grid against the oracle's statement-by-statement restatement (literal O(V^2) duplicate search in
retessellate, literal insert-at-the-front merging of the face lists).  No GPU needed."""
import numpy as np
import pytest

import newmsm_amd as M
from oracle import oracle as O


@pytest.mark.parametrize("cp_order,levels", [(0, 0), (0, 1), (1, 2), (2, 2), (1, 3), (2, 1), (3, 2)])
def test_anatomy_grid_equals_the_oracle(built, cp_order, levels):
    cxyz, ctri = M.make_mesh_from_icosa(cp_order)
    got = M.resample_anatomy_grid(cxyz, ctri, levels)
    want = O.resample_anatomy_grid(cxyz, ctri, levels, literal=True)
    for k in ("sphere_tri", "w_ptr", "w_cp", "face_ptr", "face_idx"):
        assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(got["sphere_xyz"], want["sphere_xyz"]) and np.array_equal(got["w_val"], want["w_val"])   # bit for bit
    Tc, Ta = len(ctri), len(got["sphere_tri"])
    assert Ta == Tc * 4 ** levels and np.allclose(np.linalg.norm(got["sphere_xyz"], axis=1), 100.0, atol=1e-12)
    # NEARESTFACES: every face of the anatomical sphere under exactly one control triangle, 4^levels each
    assert sorted(got["face_idx"].tolist()) == list(range(Ta)) and np.all(np.diff(got["face_ptr"]) == 4 ** levels)
    # _ANATbaryweights: three control points per vertex (ids ascending: std::map order), weights of a point inside its triangle
    assert np.all(np.diff(got["w_ptr"]) == 3)
    ids = got["w_cp"].reshape(-1, 3)
    assert np.all(np.diff(ids, axis=1) > 0) and np.allclose(got["w_val"].reshape(-1, 3).sum(axis=1), 1.0, atol=1e-12)
    assert got["w_val"].min() > -1e-9
    # the control points keep their vertex ids and sit on themselves
    w = got["w_val"].reshape(-1, 3)
    for v in range(len(cxyz)):
        assert abs(w[v][list(ids[v]).index(v)] - 1.0) < 1e-9


def test_nearest_faces_order_is_the_reference_s():
    """one pass: the children in order; every further pass puts an entry's children in FRONT of the list (insert(begin(), ...), :277-279)"""
    cxyz, ctri = M.make_mesh_from_icosa(0)
    one = M.resample_anatomy_grid(cxyz, ctri, 1)
    assert one["face_idx"][:8].tolist() == [0, 1, 2, 3, 4, 5, 6, 7]
    two = M.resample_anatomy_grid(cxyz, ctri, 2)
    assert two["face_idx"][:16].tolist() == [12, 13, 14, 15, 8, 9, 10, 11, 4, 5, 6, 7, 0, 1, 2, 3]
    three = M.resample_anatomy_grid(cxyz, ctri, 3)
    assert three["face_idx"][:8].tolist() == [12, 13, 14, 15, 8, 9, 10, 11] and three["face_idx"][56:64].tolist() == [52, 53, 54, 55, 48, 49, 50, 51]


def test_a_vertex_on_a_shared_edge_takes_the_later_triangle_s_weights():
    """baryweights[id] = ... in the loop over control triangles, their faces and corners (:303-321): the last assignment stands"""
    cxyz, ctri = M.make_mesh_from_icosa(1)
    g = M.resample_anatomy_grid(cxyz, ctri, 1)
    ids = g["w_cp"].reshape(-1, 3)
    last = {}
    for i in range(len(ctri)):
        for f in g["face_idx"][g["face_ptr"][i]:g["face_ptr"][i + 1]]:
            for a in g["sphere_tri"][f]:
                last[int(a)] = i
    for a, i in last.items():
        assert sorted(ctri[i].tolist()) == ids[a].tolist()


def test_bad_arguments_are_refused():
    cxyz, ctri = M.make_mesh_from_icosa(1)
    with pytest.raises(M.MsmError):
        M.resample_anatomy_grid(cxyz, ctri, -1)
    bad = ctri.copy()
    bad[0, 0] = len(cxyz) + 3
    with pytest.raises(M.MsmError):
        M.resample_anatomy_grid(cxyz, bad, 1)
