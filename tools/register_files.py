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

With --inanat / --refanat (both or none, CLI/newmsm.cpp:40-45) and --regoption=5 in the configuration the run is an aMSM one: the anatomical
surfaces are loaded as they are (set_anatomical, M/mesh_registration.cpp:434-438: no recentre, no rescale).

Groupwise mode (CLI/newmsm.cpp:13-27, -g / --groupwise):

    python tools/register_files.py --groupwise --meshes=mesh_list.txt --data=data_list.txt --template=template.sphere.surf.gii [--mask=mask.func.gii] \
                                   --conf=gMSM_config.txt --out=/path/prefix.

--meshes / --data: text files with one path per line (read_ascii_list, M/mesh_registration.cpp:871-884), subject i = line i of both; every mesh
and the template recentred and rescaled to RAD (M/group_mesh_registration.h:46-57,72-77).  Outputs, per subject i (M/group_mesh_registration.cpp:
120-133, .h:79-82): <out>sphere-<i>.reg<surf>, <out>sphere-<i>.LR.reg<surf>, <out>transformed_and_reprojected-<i><data> (the subject's data
resampled from its registered sphere onto the TEMPLATE).

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
from newmsm_amd import config, group_registration, meshio, registration  # noqa: E402

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
    ap.add_argument("-M", "--inmesh", default="", help="input mesh (GIFTI or FreeSurfer ASCII); needs to be a sphere")
    ap.add_argument("-R", "--refmesh", default="", help="reference mesh; the input mesh when not given")
    ap.add_argument("-i", "--indata", default="", help="scalar or multivariate data for input (.func.gii / .shape.gii / .asc / .dpv / .txt)")
    ap.add_argument("-I", "--refdata", default="", help="scalar or multivariate data for reference")
    ap.add_argument("-w", "--inweight", default="", help="cost function weighting for input")
    ap.add_argument("-W", "--refweight", default="", help="cost function weighting for reference")
    ap.add_argument("-t", "--trans", default="", help="(not supported: initialisation from a previous registration)")
    ap.add_argument("-a", "--inanat", default="", help="input anatomical mesh (the input sphere's vertices on the anatomical surface; --regoption=5)")
    ap.add_argument("-A", "--refanat", default="", help="reference anatomical mesh")
    ap.add_argument("-g", "--groupwise", action="store_true", help="run newMSM in groupwise mode")
    ap.add_argument("-m", "--meshes", default="", help="groupwise mode only; list of paths to input meshes. Needs to be a sphere")
    ap.add_argument("--template", default="", help="groupwise mode only; templates sphere for resampling. Needs to be a sphere")
    ap.add_argument("--data", default="", help="groupwise mode only; list of paths of the data files")
    ap.add_argument("--mask", default="", help="groupwise mode only; mask file path")
    ap.add_argument("-o", "--out", required=True, help="output basename")
    ap.add_argument("-f", "--format", default="GIFTI", help="format of output files: GIFTI, ASCII or ASCII_MAT")
    ap.add_argument("-c", "--conf", default="", help="configuration file")
    ap.add_argument("-v", "--verbose", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    return ap.parse_args(argv)


def read_ascii_list(path):
    """Mesh_registration::read_ascii_list, M/mesh_registration.cpp:871-884: the whitespace-separated entries of a text file"""
    with open(path) as f:
        return f.read().split()


def read_conf(path):
    if not path:
        return None
    with open(path) as f:
        return f.read()


def discrete_levels(cfg, D, anat=False, groupwise=False):
    levels, run_kw, skipped = config.levels_from_config(cfg, D, anat=anat, groupwise=groupwise)
    for index, method in skipped:
        print("register_files.py: level %d (--opt=%s) is outside the path (the affine stage stays on the CPU in newmsm): skipped" % (index + 1, method), file=sys.stderr)
    if not levels:
        raise SystemExit("register_files.py: the configuration holds no DISCRETE level")
    return levels, run_kw


def main_groupwise(a, surf_ext, data_ext):
    """CLI/newmsm.cpp:13-27 + Group_Mesh_registration (M/group_mesh_registration.h:46-82, .cpp:26-133)"""
    for flag in ("meshes", "template", "data"):
        if not getattr(a, flag):
            raise SystemExit("register_files.py: --groupwise needs --%s" % flag)
    mesh_files, data_files = read_ascii_list(a.meshes), read_ascii_list(a.data)
    if len(mesh_files) != len(data_files):
        raise SystemExit("featurespace::Initialize do not have the same number of datasets and surface meshes")  # M/featurespace.cpp:43-44
    cfg = config.parse_config(read_conf(a.conf))
    if any(m in ("RIGID", "AFFINE") for m in cfg["opt"]):
        raise SystemExit("AFFINE/RIGID registration is not supported in groupwise mode.")  # M/group_mesh_registration.cpp:29-30
    if cfg["dopt"] != "HOCR":
        raise SystemExit("Groupwise mode is only supported in the HOCR version of MSM.")  # :87
    meshes = []
    for k, path in enumerate(mesh_files):
        if a.verbose:
            print("Mesh #%d is %s" % (k, path))
        xyz, tri = meshio.load_surface(path)
        meshes.append((on_sphere(xyz), tri))
    if a.verbose:
        print("Template is " + a.template)
    txyz, ttri = meshio.load_surface(a.template)
    txyz = on_sphere(txyz)
    datas = [meshio.load_data(path, len(meshes[k][0])) for k, path in enumerate(data_files)]
    mask = meshio.load_data(a.mask, len(txyz))[0] if a.mask else None
    levels, run_kw = discrete_levels(cfg, datas[0].shape[0], groupwise=True)
    ctx = M.Context(a.device)
    if a.verbose:
        print("This is newMSM's groupwise DISCRETE path on an MI355X (msm-mi355x).\nStarting multiresolution with %d levels." % len(levels))
    regs, level_regs, energies = group_registration.run_group_multiresolution(group_registration.ProductGroupOps(ctx), meshes, datas, txyz, ttri, levels, mask=mask,
                                                                              fixnan=cfg["fixnan"], **run_kw)
    last_tri = M.make_mesh_from_icosa(levels[-1]["data_order"])[1]
    target = M.Mesh(ctx, txyz, ttri)
    for s in range(len(meshes)):
        meshio.save_surface(a.out + "sphere-%d.reg" % s + surf_ext, regs[s], meshes[s][1])            # transform
        meshio.save_surface(a.out + "sphere-%d.LR.reg" % s + surf_ext, level_regs[-1][s], last_tri)   # saveSPH_reg
        moved = M.Mesh(ctx, regs[s], meshes[s][1])
        save_data(a.out + "transformed_and_reprojected-%d" % s + data_ext, txyz, M.metric_resample(moved, datas[s], target))  # save_transformed_data
    if a.verbose:
        for k, e in enumerate(energies):
            print("level %d: energies per iteration %s" % (k + 1, [round(v, 4) for v in e]))
    ctx.close()


def main(argv):
    a = parse_args(argv)
    surf_ext, data_ext = output_formats(a.format)
    if surf_ext == ".vtk":
        raise SystemExit("register_files.py: VTK output is not written here (GIFTI, ASCII, ASCII_MAT)")
    if a.groupwise:
        return main_groupwise(a, surf_ext, data_ext)
    for flag in ("inmesh", "indata", "refdata"):
        if not getattr(a, flag):
            raise SystemExit("register_files.py: --%s is required" % flag)
    if a.trans:
        raise SystemExit("register_files.py: --trans (a previous registration as the starting point) is not wired into the level loop")
    if bool(a.inanat) != bool(a.refanat):
        raise SystemExit("Error: must supply both anatomical meshes or none")  # CLI/newmsm.cpp:41-43
    ixyz, itri = meshio.load_surface(a.inmesh)
    rxyz, rtri = meshio.load_surface(a.refmesh or a.inmesh)
    ixyz, rxyz = on_sphere(ixyz), on_sphere(rxyz)
    idata, rdata = meshio.load_data(a.indata, len(ixyz)), meshio.load_data(a.refdata, len(rxyz))
    if idata.shape[0] != rdata.shape[0]:
        raise SystemExit("Mesh_registration: input and reference data have different numbers of feature rows (%d, %d)" % (idata.shape[0], rdata.shape[0]))
    levels, run_kw = discrete_levels(config.parse_config(read_conf(a.conf)), idata.shape[0], anat=bool(a.inanat))
    cfw = {}
    if a.inanat:  # set_anatomical: loaded as they are
        cfw.update(in_anat=meshio.load_surface(a.inanat)[0], ref_anat=meshio.load_surface(a.refanat)[0])
    if a.inweight and a.refweight:
        cfw.update(in_cfweight=meshio.load_data(a.inweight, len(ixyz)), ref_cfweight=meshio.load_data(a.refweight, len(rxyz)))
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
    try:
        main(sys.argv[1:])
    except (config.ConfigError, ValueError) as e:  # MeshregException: the message, exit status 1 (CLI/newmsm.cpp:62-65)
        raise SystemExit(str(e))
