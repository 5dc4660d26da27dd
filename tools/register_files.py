"""`newmsm` for DISCRETE levels, from files to files under newmsm's own flag names (CLI/msmOptions.h:59-157), over the MI355X path:

    python tools/register_files.py --inmesh=in.sphere.surf.gii --refmesh=ref.sphere.surf.gii --indata=in.func.gii --refdata=ref.func.gii \\
                                   --conf=config/basic_configs/config_standard_MSM_strain --out=/path/prefix. [-f GIFTI|ASCII|ASCII_MAT] [--verbose]

What CLI/newmsm.cpp:29-58 does for a pairwise run: set_input / set_reference (load, recentre, true_rescale to RAD = 100:
M/mesh_registration.cpp:416-438), the configuration file through the reference's grammar (newmsm_amd/config.py = parse_reg_options :459-784),
run_multiresolutions (:30-50: per DISCRETE level featurespace + project_CPgrid + run_discrete_opt with the optimiser --dopt names), then the three
outputs of :47-49:
    <out>sphere.reg<surf>                     transform (:352-356): the input sphere moved through the final warp
    <out>sphere.LR.reg<surf>                  saveSPH_reg (M/mesh_registration.h:170): the last level's data grid at its registered position
    <out>transformed_and_reprojected<data>    save_transformed_data (:358-408): the input data resampled from the registered sphere onto the reference
with <surf> / <data> = .surf.gii / .func.gii (GIFTI), .asc / .dpv (ASCII), .asc / .txt (ASCII_MAT) as set_output_format (:827-842) names them.

Outside the path and reported instead of silently dropped: AFFINE / RIGID levels (skipped with a note on stderr), --trans, --IN / --INc / --excl; the
binary solve of --dopt=HOCR / FastPD is a stand-in (iterated conditional modes: FastPD and ELC are licence-restricted and FSL-bound), so a run
exercises the path exactly as newmsm would but its labelings are not HOCR's.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import newmsm_amd as M  # noqa: E402
from newmsm_amd import config, meshio, registration  # noqa: E402

RAD = 100.0


def on_sphere(xyz, rad=RAD):
    """recentre + true_rescale (R/mesh.cpp:1198-1255), as Mesh_registration::set_input / set_reference apply them"""
    xyz = xyz - xyz.mean(axis=0)
    return xyz * (rad / np.linalg.norm(xyz, axis=1, keepdims=True))


def output_formats(fmt):
    """Mesh_registration::set_output_format, M/mesh_registration.cpp:827-842"""
    if fmt == "GIFTI":
        return ".surf.gii", ".func.gii"
    if fmt in ("ASCII", "ASCII_MAT"):
        return ".asc", (".dpv" if fmt == "ASCII" else ".txt")
    return ".vtk", ".txt"


def save_data(path, mesh_xyz, data):
    if path.endswith(".dpv"):
        meshio.save_dpv(path, mesh_xyz, data)
    elif path.endswith(".txt"):
        meshio.save_matrix(path, data)
    else:
        meshio.save_metric(path, data)


def parse_args(argv):
    ap = argparse.ArgumentParser(prog="register_files.py", description="newmsm's pairwise mode over the MI355X path (DISCRETE levels)")
    ap.add_argument("-M", "--inmesh", required=True, help="input mesh (GIFTI or FreeSurfer ASCII); needs to be a sphere")
    ap.add_argument("-R", "--refmesh", default="", help="reference mesh; the input mesh when not given")
    ap.add_argument("-i", "--indata", required=True, help="scalar or multivariate data for input (.func.gii / .shape.gii / .asc / .dpv / .txt)")
    ap.add_argument("-I", "--refdata", required=True, help="scalar or multivariate data for reference")
    ap.add_argument("-w", "--inweight", default="", help="cost function weighting for input")
    ap.add_argument("-W", "--refweight", default="", help="cost function weighting for reference")
    ap.add_argument("-t", "--trans", default="", help="(not supported: initialisation from a previous registration)")
    ap.add_argument("-a", "--inanat", default="", help="(not supported here: anatomical meshes of --regoption=5)")
    ap.add_argument("-A", "--refanat", default="")
    ap.add_argument("-o", "--out", required=True, help="output basename")
    ap.add_argument("-f", "--format", default="GIFTI", help="format of output files: GIFTI, ASCII or ASCII_MAT")
    ap.add_argument("-c", "--conf", default="", help="configuration file")
    ap.add_argument("-v", "--verbose", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    return ap.parse_args(argv)


def main(argv):
    a = parse_args(argv)
    if a.trans:
        raise SystemExit("register_files.py: --trans (a previous registration as the starting point) is not wired into the level loop")
    if bool(a.inanat) != bool(a.refanat):
        raise SystemExit("Error: must supply both anatomical meshes or none")  # CLI/newmsm.cpp:41-43
    if a.inanat:
        raise SystemExit("register_files.py: anatomical meshes (--regoption=5) are not read here")
    surf_ext, data_ext = output_formats(a.format)
    if surf_ext == ".vtk":
        raise SystemExit("register_files.py: VTK output is not written here (GIFTI, ASCII, ASCII_MAT)")
    ixyz, itri = meshio.load_surface(a.inmesh)
    rxyz, rtri = meshio.load_surface(a.refmesh or a.inmesh)
    ixyz, rxyz = on_sphere(ixyz), on_sphere(rxyz)
    idata, rdata = meshio.load_data(a.indata, len(ixyz)), meshio.load_data(a.refdata, len(rxyz))
    if idata.shape[0] != rdata.shape[0]:
        raise SystemExit("Mesh_registration: input and reference data have different numbers of feature rows (%d, %d)" % (idata.shape[0], rdata.shape[0]))
    text = None
    if a.conf:
        with open(a.conf) as f:
            text = f.read()
    levels, run_kw, skipped = config.levels_from_config(config.parse_config(text), idata.shape[0])
    for index, method in skipped:
        print("register_files.py: level %d (--opt=%s) is outside the path (the affine stage stays on the CPU in newmsm): skipped" % (index + 1, method), file=sys.stderr)
    if not levels:
        raise SystemExit("register_files.py: the configuration holds no DISCRETE level")
    cfw = {}
    if a.inweight and a.refweight:
        cfw = dict(in_cfweight=meshio.load_data(a.inweight, len(ixyz)), ref_cfweight=meshio.load_data(a.refweight, len(rxyz)))
    ctx = M.Context(a.device)
    if a.verbose:
        print("This is newMSM's DISCRETE path on an MI355X (msm-mi355x).\nStarting multiresolution with %d levels." % len(levels))
    reg, level_regs, energies = registration.run_multiresolution(registration.ProductOps(ctx), ixyz, itri, idata, rxyz, rtri, rdata, levels, **run_kw, **cfw)
    out = a.out
    meshio.save_surface(out + "sphere.reg" + surf_ext, reg, itri)                                   # transform
    last_xyz, last_tri = M.make_mesh_from_icosa(levels[-1]["data_order"])
    meshio.save_surface(out + "sphere.LR.reg" + surf_ext, level_regs[-1], last_tri)                 # saveSPH_reg
    moved, target = M.Mesh(ctx, reg, itri), M.Mesh(ctx, rxyz, rtri)
    save_data(out + "transformed_and_reprojected" + data_ext, rxyz, M.metric_resample(moved, idata, target))  # save_transformed_data
    if a.verbose:
        for k, e in enumerate(energies):
            print("level %d: energies per iteration %s" % (k + 1, [round(v, 4) for v in e]))
    ctx.close()


if __name__ == "__main__":
    main(sys.argv[1:])
