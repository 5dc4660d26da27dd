"""include/msmhip_io.hpp (the C++ host's GIFTI / FreeSurfer-ASCII reader and writer, R/mesh.cpp:350-398,455-515,582-631) against
newmsm_amd/meshio.py: the two writers produce the same bytes, each reads what the other wrote, and the C++ reader decodes the
hand-written known-answer files of tests/test_meshio.py in every encoding of the GIFTI 1.0 specification.  No GPU needed."""
import os
import subprocess

import numpy as np
import pytest

from newmsm_amd import meshio, synthetic
from tests.test_meshio import HEAD, data_array

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "io_roundtrip.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "io_roundtrip")


@pytest.fixture(scope="module")
def exe():
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE, "-lz", "-lexpat"])
    return EXE


def run(exe, mode, src, dst):
    r = subprocess.run([exe, mode, str(src), str(dst)], capture_output=True, text=True, timeout=120)
    return r.returncode, r.stderr


def icosahedron():
    t = (1 + 5 ** 0.5) / 2
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8], [3, 9, 4], [3, 4, 2],
                  [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int32)
    return v / np.linalg.norm(v, axis=1, keepdims=True) * 100.0, f


def test_surface_written_by_either_side_is_the_same_file(exe, tmp_path):
    xyz, tri = icosahedron()
    xyz = xyz * (1 + 1e-3 * np.sin(np.arange(12)))[:, None]  # not exactly representable in float32
    py, cpp = tmp_path / "py.surf.gii", tmp_path / "cpp.surf.gii"
    meshio.save_surface(str(py), xyz, tri)
    assert run(exe, "surf", py, cpp)[0] == 0
    assert py.read_bytes() == cpp.read_bytes()  # same XML, same deflate stream, same base64
    gx, gt = meshio.load_surface(str(cpp))
    assert np.array_equal(gx, xyz.astype(np.float32).astype(np.float64)) and np.array_equal(gt, tri)


def test_metric_round_trip_and_bytes(exe, tmp_path):
    rng = np.random.default_rng(3)
    data = rng.normal(size=(5, 642))
    py, cpp = tmp_path / "py.func.gii", tmp_path / "cpp.func.gii"
    meshio.save_metric(str(py), data)
    assert run(exe, "metric", py, cpp)[0] == 0
    assert py.read_bytes() == cpp.read_bytes()
    assert np.array_equal(meshio.load_metric(str(cpp)), data.astype(np.float32).astype(np.float64))


def test_reader_decodes_every_encoding(exe, tmp_path):
    import base64
    import struct
    import zlib

    xyz = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0], [0.5, 0.25, -1.5]], dtype=np.float32)
    tri = np.array([[0, 1, 2], [1, 3, 2]], dtype=np.int32)
    b64 = lambda a, fmt: base64.b64encode(struct.pack(fmt, *a.ravel().tolist())).decode()
    files = {
        "ascii": (data_array("NIFTI_INTENT_POINTSET", "NIFTI_TYPE_FLOAT32", (4, 3), "ASCII", " ".join("%g" % v for v in xyz.ravel()))
                  + data_array("NIFTI_INTENT_TRIANGLE", "NIFTI_TYPE_INT32", (2, 3), "ASCII", "0 1 2\n1 3 2")),
        "b64_big_endian": (data_array("NIFTI_INTENT_POINTSET", "NIFTI_TYPE_FLOAT32", (4, 3), "Base64Binary", b64(xyz, ">12f"), endian="BigEndian")
                           + data_array("NIFTI_INTENT_TRIANGLE", "NIFTI_TYPE_INT32", (2, 3), "Base64Binary", b64(tri, ">6i"), endian="BigEndian")),
        "gz_column_major": (data_array("NIFTI_INTENT_POINTSET", "NIFTI_TYPE_FLOAT32", (4, 3), "GZipBase64Binary",
                                       base64.b64encode(zlib.compress(np.asfortranarray(xyz).tobytes(order="F"))).decode(), order="ColumnMajorOrder")
                            + data_array("NIFTI_INTENT_TRIANGLE", "NIFTI_TYPE_INT32", (2, 3), "GZipBase64Binary", base64.b64encode(zlib.compress(tri.tobytes())).decode())),
        "int16_f64": (data_array("NIFTI_INTENT_POINTSET", "NIFTI_TYPE_FLOAT64", (4, 3), "Base64Binary", b64(xyz.astype(np.float64), "<12d"))
                      + data_array("NIFTI_INTENT_TRIANGLE", "NIFTI_TYPE_INT16", (2, 3), "Base64Binary", b64(tri, "<6h"))),
    }
    for name, body in files.items():
        p, out = tmp_path / (name + ".surf.gii"), tmp_path / (name + ".txt")
        p.write_text(HEAD % 2 + body + "</GIFTI>\n")
        assert run(exe, "dump", p, out)[0] == 0, name
        lines = out.read_text().splitlines()
        assert lines[0] == "NIFTI_INTENT_POINTSET 4 3" and lines[2] == "NIFTI_INTENT_TRIANGLE 2 3", name
        assert np.array_equal(np.array(lines[1].split(), dtype=np.float64), xyz.astype(np.float64).ravel()), name
        assert np.array_equal(np.array(lines[3].split(), dtype=np.float64), tri.ravel()), name
        gx, gt = meshio.load_surface(str(p))  # and the Python reader agrees
        assert np.array_equal(gx, xyz.astype(np.float64)) and np.array_equal(gt, tri)


def test_freesurfer_ascii_both_ways(exe, tmp_path):
    xyz, tri = icosahedron()
    val = synthetic.smooth_feature(xyz, 0, 5)
    py, cpp = tmp_path / "py.asc", tmp_path / "cpp.asc"
    meshio.save_ascii(str(py), xyz, tri, None)
    assert run(exe, "surf", py, cpp)[0] == 0
    assert py.read_bytes() == cpp.read_bytes()
    meshio.save_ascii(str(py), xyz, tri, val)
    m = tmp_path / "v.func.gii"
    assert run(exe, "metric", py, m)[0] == 0  # the value column of an .asc file as a metric
    assert np.array_equal(meshio.load_metric(str(m))[0], val.astype(np.float32).astype(np.float64))


def test_reader_errors(exe, tmp_path):
    bad = tmp_path / "bad.surf.gii"
    bad.write_text(HEAD % 1 + data_array("NIFTI_INTENT_POINTSET", "NIFTI_TYPE_FLOAT32", (2, 3), "ASCII", "1 2 3") + "</GIFTI>\n")
    rc, err = run(exe, "surf", bad, tmp_path / "o.gii")
    assert rc == 1 and "dimensions say 6" in err
    bad.write_text("<NotGifti/>")
    rc, err = run(exe, "surf", bad, tmp_path / "o.gii")
    assert rc == 1 and "root element" in err
    bad.write_text(HEAD % 1 + data_array("NIFTI_INTENT_POINTSET", "NIFTI_TYPE_FLOAT32", (1, 3), "ASCII", "1 2 3") + "</GIFTI>\n")
    rc, err = run(exe, "surf", bad, tmp_path / "o.gii")
    assert rc == 1 and "holds no surface" in err
