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
