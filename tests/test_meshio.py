"""GIFTI / FreeSurfer-ASCII files on either side of the path (newmsm_amd/meshio.py): known-answer files written by hand in
the encodings of the GIFTI 1.0 specification, and round trips with the float32 rounding the reference applies on save."""
import base64
import struct
import zlib

import numpy as np
import pytest

from newmsm_amd import meshio

HEAD = '<?xml version="1.0" encoding="UTF-8"?>\n<GIFTI Version="1.0" NumberOfDataArrays="%d">\n<MetaData/>\n<LabelTable/>\n'


def data_array(intent, dtype, dims, encoding, payload, order="RowMajorOrder", endian="LittleEndian"):
    d = " ".join('Dim%d="%d"' % (k, n) for k, n in enumerate(dims))
    return ('<DataArray Intent="%s" DataType="%s" ArrayIndexingOrder="%s" Dimensionality="%d" %s Encoding="%s" Endian="%s" '
            'ExternalFileName="" ExternalFileOffset="">\n<MetaData/>\n<Data>%s</Data>\n</DataArray>\n' % (intent, dtype, order, len(dims), d, encoding, endian, payload))


def test_known_answer_surface_in_three_encodings(tmp_path):
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
                            + data_array("NIFTI_INTENT_TRIANGLE", "NIFTI_TYPE_INT32", (2, 3), "GZipBase64Binary",
                                         base64.b64encode(zlib.compress(tri.tobytes())).decode())),
    }
    for name, body in files.items():
        p = tmp_path / (name + ".surf.gii")
        p.write_text(HEAD % 2 + body + "</GIFTI>\n")
        gx, gt = meshio.load_surface(str(p))
        assert gx.dtype == np.float64 and gt.dtype == np.int32
        assert np.array_equal(gx, xyz.astype(np.float64)) and np.array_equal(gt, tri), name


def test_surface_and_metric_round_trip(tmp_path, built):
    import newmsm_amd as M

    xyz, tri = M.make_mesh_from_icosa(3)
    p = str(tmp_path / "sphere.surf.gii")
    meshio.save_surface(p, xyz, tri)
    gx, gt = meshio.load_surface(p)
    assert np.array_equal(gt, tri)
    assert np.array_equal(gx, xyz.astype(np.float32).astype(np.float64))  # surfaces are written float32 (R/mesh.cpp:607-609)
    text = open(p).read()
    assert 'Encoding="GZipBase64Binary"' in text and 'Intent="NIFTI_INTENT_POINTSET"' in text and 'DataType="NIFTI_TYPE_INT32"' in text
    data = np.random.default_rng(0).normal(size=(3, len(xyz)))
    q = str(tmp_path / "data.func.gii")
    meshio.save_metric(q, data)
    got = meshio.load_metric(q, nvertices=len(xyz))
    assert got.shape == (3, len(xyz)) and np.array_equal(got, data.astype(np.float32).astype(np.float64))
    with pytest.raises(meshio.MeshIOError, match="mismatch between data and surface dimensions"):
        meshio.load_metric(q, nvertices=len(xyz) + 1)
    with pytest.raises(meshio.MeshIOError, match="no surface"):
        meshio.load_surface(q)


def test_multi_column_metric_and_bad_files(tmp_path):
    vals = np.arange(8, dtype=np.float32).reshape(4, 2)
    p = tmp_path / "two.func.gii"
    p.write_text(HEAD % 1 + data_array("NIFTI_INTENT_NONE", "NIFTI_TYPE_FLOAT32", (4, 2), "ASCII", " ".join(str(v) for v in vals.ravel())) + "</GIFTI>\n")
    assert np.array_equal(meshio.load_metric(str(p)), vals.T.astype(np.float64))
    bad = tmp_path / "bad.surf.gii"
    bad.write_text(HEAD % 1 + data_array("NIFTI_INTENT_POINTSET", "NIFTI_TYPE_FLOAT32", (4, 3), "ASCII", "1 2 3") + "</GIFTI>\n")
    with pytest.raises(meshio.MeshIOError, match="holds 3 values"):
        meshio.load_surface(str(bad))
    bad.write_text("<NotGifti/>")
    with pytest.raises(meshio.MeshIOError, match="root element"):
        meshio.read_gifti(str(bad))


def test_freesurfer_ascii(tmp_path):
    p = tmp_path / "tiny.asc"
    p.write_text("#!ascii version of tiny\n4 2\n0 0 1 0.5\n1 0 0 1.5\n0 1 0 -2\n0.5 0.25 -1.5 3.25\n0 1 2 0\n1 3 2 0\n")
    xyz, tri = meshio.load_surface(str(p))
    assert np.array_equal(xyz, [[0, 0, 1], [1, 0, 0], [0, 1, 0], [0.5, 0.25, -1.5]]) and np.array_equal(tri, [[0, 1, 2], [1, 3, 2]])
    assert np.array_equal(meshio.load_metric(str(p)), [[0.5, 1.5, -2.0, 3.25]])
    q = str(tmp_path / "out.asc")
    meshio.save_ascii(q, xyz * 1.1, tri, values=[1, 2, 3, 4])
    x2, t2 = meshio.load_surface(q)
    assert np.array_equal(x2, xyz * 1.1) and np.array_equal(t2, tri) and np.array_equal(meshio.load_metric(q), [[1, 2, 3, 4]])
    p.write_text("no header\n")
    with pytest.raises(meshio.MeshIOError, match="error in the header"):
        meshio.load_surface(str(p))


def test_dpv_and_matrix_text_files(tmp_path):
    """the ASCII / ASCII_MAT data formats of set_output_format (M/mesh_registration.cpp:827-842): Mesh::save_dpv (R/mesh.cpp:707-741: `index x y z value`,
    indices below 100 padded to three digits, six significant digits as std::ostream writes floats) and Mesh::save_matrix (:743-766: a line per data row)"""
    rng = np.random.default_rng(0)
    xyz, val = rng.normal(size=(130, 3)) * 100.0, rng.normal(size=(2, 130))
    p = str(tmp_path / "a.dpv")
    meshio.save_dpv(p, xyz, val)
    lines = open(p).read().splitlines()
    assert len(lines) == 130 and lines[7].split()[0] == "007" and lines[99].split()[0] == "099" and lines[100].split()[0] == "100"
    assert lines[3] == "003 " + " ".join("%g" % float(np.float32(v)) for v in (*xyz[3], val[0, 3]))
    x2, v2 = meshio.load_dpv(p)
    assert np.allclose(x2, xyz, rtol=1e-5, atol=1e-4) and np.allclose(v2[0], val[0], rtol=1e-5, atol=1e-6)
    q = str(tmp_path / "m.txt")
    meshio.save_matrix(q, val)
    rows = open(q).read().splitlines()
    assert len(rows) == 2 and rows[0].endswith(" ") and len(rows[0].split()) == 130
    assert np.allclose(meshio.load_data(q, 130), val, rtol=1e-5, atol=1e-6)
    assert meshio.load_data(p, 130).shape == (1, 130)
