"""Test-side glue: builds ORACLE objects (oracle/ is test infrastructure) from the numpy inputs that
newmsm_amd.problem prepares for the product path."""
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
    """Anatomical-regulariser inputs (control grid, anatomical sphere, _ANATbaryweights, NEARESTFACES) built with the oracle
    only, shaped like Mesh_registration::resample_anatomy (M/mesh_registration.cpp:250-332) prepares them."""
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


