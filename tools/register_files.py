"""File in, file out: registers a moving sphere to a reference sphere from GIFTI (or FreeSurfer ASCII) files with the loops of
newmsm_amd/registration.py, and writes what newMSM's `transform` / `save_transformed_data` write (M/mesh_registration.cpp:
352-408): <out>sphere.reg.surf.gii and <out>transformed_and_reprojected.func.gii.  Not a replacement of the `newmsm` CLI (no
config parser, Monte Carlo optimiser only) -- an example of the data formats either side of the path.

    python tools/register_files.py in.sphere.surf.gii ref.sphere.surf.gii in.func.gii ref.func.gii out_prefix [iters mciters]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
import newmsm_amd as M  # noqa: E402
from newmsm_amd import meshio, registration  # noqa: E402


def on_sphere(xyz, rad=100.0):  # recentre + true_rescale, M/mesh_registration.cpp:416-457
    xyz = xyz - xyz.mean(axis=0)
    return xyz * (rad / np.linalg.norm(xyz, axis=1, keepdims=True))


def main(argv):
    in_surf, ref_surf, in_data, ref_data, out = argv[:5]
    iters, mciters = (int(a) for a in (argv[5:7] + ["3", "200"][len(argv) - 5:]))
    ixyz, itri = meshio.load_surface(in_surf)
    rxyz, rtri = meshio.load_surface(ref_surf)
    ixyz, rxyz = on_sphere(ixyz), on_sphere(rxyz)
    idata, rdata = meshio.load_metric(in_data, len(ixyz)), meshio.load_metric(ref_data, len(rxyz))
    levels = [dict(data_order=4, cp_order=2, sigma_in=4.0, sigma_ref=4.0), dict(data_order=5, cp_order=3, sigma_in=2.0, sigma_ref=2.0),
              dict(data_order=6, cp_order=4, sigma_in=1.0, sigma_ref=1.0)]
    ctx = M.Context(0)
    kind = "multivariate" if idata.shape[0] > 1 else "univariate"
    reg, _, energies = registration.run_multiresolution(registration.ProductOps(ctx), ixyz, itri, idata, rxyz, rtri, rdata, levels, varnorm=True,
                                                        iters=iters, mciters=mciters, mcparam=0.8, seed=0, kind=kind)
    meshio.save_surface(out + "sphere.reg.surf.gii", reg, itri)
    moved, target = M.Mesh(ctx, reg, itri), M.Mesh(ctx, rxyz, rtri)
    meshio.save_metric(out + "transformed_and_reprojected.func.gii", M.metric_resample(moved, idata, target))
    print("energies per level:", [[round(e, 3) for e in lv] for lv in energies])


if __name__ == "__main__":
    if len(sys.argv) < 6:
        sys.exit(__doc__)
    main(sys.argv[1:])
