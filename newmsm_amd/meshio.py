"""Mesh and metric files on either side of the path, in the two formats newMSM's tools exchange:

* GIFTI (`.surf.gii`, `.func.gii`, `.shape.gii`) as `Mesh::load_gifti` / `Mesh::save_gifti` read and write it
  (R/mesh.cpp:350-398, 582-631): a surface is a NIFTI_INTENT_POINTSET array (float32, N x 3) followed by a
  NIFTI_INTENT_TRIANGLE array (int32, T x 3); a metric file holds one float32 array of N values per feature; files are
  written GZipBase64Binary, row-major, little-endian.  The reader also accepts ASCII and Base64Binary encodings, big-endian
  data, column-major arrays and the other numeric GIFTI datatypes.
* FreeSurfer ASCII (`.asc`) as `Mesh::load_ascii` reads it (R/mesh.cpp:455-515): "#!ascii" header, "NVertices NFaces", then
  "x y z value" per vertex and "a b c value" per face.

newMSM reads GIFTI through FSL's giftiInterface, which is not part of the reference tree; this module follows the GIFTI 1.0
specification and the reference's call sites.  Host-side only (numpy + the standard library); coordinates are returned
as float64 arrays holding the float32 values of the file, which is what Mpoint receives in the reference.
"""
import base64
import sys
import xml.etree.ElementTree as ET
import zlib

import numpy as np

_DTYPES = {
    "NIFTI_TYPE_UINT8": np.uint8, "NIFTI_TYPE_INT8": np.int8, "NIFTI_TYPE_INT16": np.int16, "NIFTI_TYPE_UINT16": np.uint16,
    "NIFTI_TYPE_INT32": np.int32, "NIFTI_TYPE_UINT32": np.uint32, "NIFTI_TYPE_INT64": np.int64, "NIFTI_TYPE_UINT64": np.uint64,
    "NIFTI_TYPE_FLOAT32": np.float32, "NIFTI_TYPE_FLOAT64": np.float64,
}
_IDENTITY = "1.000000 0.000000 0.000000 0.000000 0.000000 1.000000 0.000000 0.000000 0.000000 0.000000 1.000000 0.000000 0.000000 0.000000 0.000000 1.000000"


class MeshIOError(ValueError):
    pass


# ------------------------------------------------------------------------------------------------ GIFTI
def _decode_array(da):
    dtype = _DTYPES.get(da.get("DataType"))
    if dtype is None:
        raise MeshIOError("GIFTI: unsupported DataType %r" % da.get("DataType"))
    ndim = int(da.get("Dimensionality", "1"))
    dims = [int(da.get("Dim%d" % k)) for k in range(ndim)]
    count = int(np.prod(dims)) if dims else 0
    enc = da.get("Encoding", "ASCII")
    data = da.find("Data")
    text = (data.text or "") if data is not None else ""
    if enc == "ASCII":
        flat = np.array(text.split(), dtype=np.float64).astype(dtype) if dtype in (np.float32, np.float64) else np.array(text.split(), dtype=np.int64).astype(dtype)
    elif enc in ("Base64Binary", "GZipBase64Binary"):
        raw = base64.b64decode(text.strip())
        if enc == "GZipBase64Binary":
            raw = zlib.decompress(raw)
        order = "<" if da.get("Endian", "LittleEndian") == "LittleEndian" else ">"
        flat = np.frombuffer(raw, dtype=np.dtype(dtype).newbyteorder(order)).astype(dtype)
    else:
        raise MeshIOError("GIFTI: Encoding %r is not supported (external files are not read)" % enc)
    if flat.size != count:
        raise MeshIOError("GIFTI: array holds %d values, its dimensions say %d" % (flat.size, count))
    if da.get("ArrayIndexingOrder", "RowMajorOrder") == "ColumnMajorOrder" and ndim > 1:
        return np.ascontiguousarray(flat.reshape(dims[::-1]).T)
    return flat.reshape(dims)


def read_gifti(path):
    """All data arrays of a GIFTI file as (intent, ndarray) pairs, in file order."""
    try:
        root = ET.parse(path).getroot()
    except ET.ParseError as e:
        raise MeshIOError("GIFTI: %s is not well-formed XML (%s)" % (path, e))
    if root.tag != "GIFTI":
        raise MeshIOError("GIFTI: %s has root element <%s>" % (path, root.tag))
    return [(da.get("Intent", "NIFTI_INTENT_NONE"), _decode_array(da)) for da in root.findall("DataArray")]


def _encode_array(intent, a, with_coordsys=False):
    a = np.ascontiguousarray(a)
    dt = {np.dtype(np.float32): "NIFTI_TYPE_FLOAT32", np.dtype(np.int32): "NIFTI_TYPE_INT32"}[a.dtype]
    attrs = ['Intent="%s"' % intent, 'DataType="%s"' % dt, 'ArrayIndexingOrder="RowMajorOrder"', 'Dimensionality="%d"' % a.ndim]
    attrs += ['Dim%d="%d"' % (k, n) for k, n in enumerate(a.shape)]
    attrs += ['Encoding="GZipBase64Binary"', 'Endian="LittleEndian"', 'ExternalFileName=""', 'ExternalFileOffset=""']
    payload = base64.b64encode(zlib.compress(a.astype(a.dtype.newbyteorder("<")).tobytes())).decode("ascii")
    out = ["   <DataArray %s>" % " ".join(attrs), "      <MetaData/>"]
    if with_coordsys:
        out += ["      <CoordinateSystemTransformMatrix>", "         <DataSpace><![CDATA[NIFTI_XFORM_UNKNOWN]]></DataSpace>",
                "         <TransformedSpace><![CDATA[NIFTI_XFORM_UNKNOWN]]></TransformedSpace>",
                "         <MatrixData>%s</MatrixData>" % _IDENTITY, "      </CoordinateSystemTransformMatrix>"]
    out += ["      <Data>%s</Data>" % payload, "   </DataArray>"]
    return "\n".join(out)


def _write_gifti(path, arrays):
    body = ['<?xml version="1.0" encoding="UTF-8"?>', '<!DOCTYPE GIFTI SYSTEM "http://www.nitrc.org/frs/download.php/115/gifti.dtd">',
            '<GIFTI Version="1.0" NumberOfDataArrays="%d">' % len(arrays), "   <MetaData/>", "   <LabelTable/>"]
    body += arrays + ["</GIFTI>", ""]
    with open(path, "w") as f:
        f.write("\n".join(body))


def load_surface(path):
    """(xyz [V x 3 float64], tri [T x 3 int32]) of a .surf.gii or FreeSurfer .asc file."""
    if str(path).endswith(".asc"):
        xyz, tri, _ = _read_ascii(path)
        return xyz, tri
    arrays = read_gifti(path)
    pts = [a for i, a in arrays if i == "NIFTI_INTENT_POINTSET"]
    tris = [a for i, a in arrays if i == "NIFTI_INTENT_TRIANGLE"]
    if not pts or not tris:
        raise MeshIOError("GIFTI: %s holds no surface (POINTSET + TRIANGLE arrays)" % path)
    xyz, tri = pts[0], tris[0]
    if xyz.ndim != 2 or xyz.shape[1] != 3 or tri.ndim != 2 or tri.shape[1] != 3:
        raise MeshIOError("GIFTI: surface arrays must be N x 3")
    tri = tri.astype(np.int32)
    if tri.size and (tri.min() < 0 or tri.max() >= len(xyz)):
        raise MeshIOError("GIFTI: triangle refers to a vertex that does not exist")
    return xyz.astype(np.float64), tri


def save_surface(path, xyz, tri):
    """save_gifti for a '.surf' file: float32 coordinates (as the reference writes them), int32 triangles."""
    if str(path).endswith(".asc"):
        return _write_ascii(path, xyz, tri, None)
    xyz = np.asarray(xyz, dtype=np.float32).reshape(-1, 3)
    tri = np.asarray(tri, dtype=np.int32).reshape(-1, 3)
    _write_gifti(path, [_encode_array("NIFTI_INTENT_POINTSET", xyz, True), _encode_array("NIFTI_INTENT_TRIANGLE", tri, True)])


def load_metric(path, nvertices=None):
    """D x V float64 matrix of a .func.gii / .shape.gii (one array per feature) or of the value column of an .asc file."""
    if str(path).endswith(".asc"):
        _, _, val = _read_ascii(path)
        return val[None, :]
    rows = []
    for intent, a in read_gifti(path):
        if intent in ("NIFTI_INTENT_POINTSET", "NIFTI_INTENT_TRIANGLE"):
            continue
        a = a.reshape(a.shape[0], -1)
        if nvertices is not None and a.shape[0] != nvertices:
            raise MeshIOError(" mismatch between data and surface dimensions")  # R/mesh.cpp:392
        rows += [a[:, k].astype(np.float64) for k in range(a.shape[1])]
    if not rows:
        raise MeshIOError("GIFTI: %s holds no data arrays" % path)
    if len({len(r) for r in rows}) != 1:
        raise MeshIOError(" mismatch between data and surface dimensions")
    return np.stack(rows)


def save_metric(path, data):
    """save_gifti for a '.func' / '.shape' file: one float32 NIFTI_INTENT_NONE array per feature row."""
    data = np.atleast_2d(np.asarray(data, dtype=np.float32))
    _write_gifti(path, [_encode_array("NIFTI_INTENT_NONE", row) for row in data])


# ------------------------------------------------------------------------------------------------ FreeSurfer ASCII
def _read_ascii(path):
    with open(path) as f:
        header = f.readline()
        if "#!ascii" not in header:
            raise MeshIOError("Mesh::load_ascii:error in the header")
        tok = f.read().split()
    try:
        nv, nf = int(tok[0]), int(tok[1])
        v = np.array(tok[2:2 + 4 * nv], dtype=np.float64).reshape(nv, 4)
        t = np.array(tok[2 + 4 * nv:2 + 4 * nv + 4 * nf], dtype=np.float64).reshape(nf, 4)
    except (ValueError, IndexError):
        raise MeshIOError("Mesh::load_ascii: %s is truncated" % path)
    return v[:, :3].copy(), t[:, :3].astype(np.int32), v[:, 3].astype(np.float32).astype(np.float64)  # values pass through a float (:486)


def _write_ascii(path, xyz, tri, values):
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    tri = np.asarray(tri, dtype=np.int32).reshape(-1, 3)
    val = np.zeros(len(xyz)) if values is None else np.asarray(values, dtype=np.float64)
    with open(path, "w") as f:
        f.write("#!ascii from msm-mi355x\n%d %d\n" % (len(xyz), len(tri)))
        for p, v in zip(xyz, val):
            f.write("%.17g %.17g %.17g %.9g\n" % (p[0], p[1], p[2], v))
        for t in tri:
            f.write("%d %d %d 0\n" % (t[0], t[1], t[2]))


def save_ascii(path, xyz, tri, values=None):
    _write_ascii(path, xyz, tri, values)


def _g(x):
    """a float as std::ostream writes it by default (6 significant digits, %g)"""
    return "%g" % float(np.float32(x))


def save_dpv(path, xyz, values):
    """Mesh::save_dpv, R/mesh.cpp:707-741: `index x y z value` per vertex (indices below 100 zero-padded to three digits), first data row only"""
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    val = np.atleast_2d(np.asarray(values, dtype=np.float64))[0]
    if len(val) != len(xyz):
        raise MeshIOError("Mesh::save_dpv, data and mesh dimensions do not agree")
    with open(path, "w") as f:
        for i, (p, v) in enumerate(zip(xyz, val)):
            f.write("%s %s %s %s %s\n" % ("%03d" % i if i < 100 else str(i), _g(p[0]), _g(p[1]), _g(p[2]), _g(v)))


def load_dpv(path):
    """(xyz, values 1 x V) of a .dpv file (Mesh::load_ascii_file, R/mesh.cpp:517-549: five columns)"""
    a = np.loadtxt(path, dtype=np.float64, ndmin=2)
    if a.shape[1] != 5:
        raise MeshIOError("Mesh::load_dpv:error opening file (wrong format) : %s" % path)
    return a[:, 1:4].copy(), a[:, 4][None, :].copy()


def save_matrix(path, data):
    """Mesh::save_matrix, R/mesh.cpp:743-766: one line per data row, values separated (and followed) by a blank"""
    with open(path, "w") as f:
        for row in np.atleast_2d(np.asarray(data, dtype=np.float64)):
            f.write("".join(_g(v) + " " for v in row) + "\n")


def load_matrix(path):
    return np.atleast_2d(np.loadtxt(path, dtype=np.float64, ndmin=2))


def load_data(path, nvertices=None):
    """the data of a --indata / --refdata file by its extension, D x V (set_data, M/reg_tools.cpp:846-867 / Mesh::load, R/mesh.cpp:296-348)"""
    p = str(path)
    if p.endswith(".dpv"):
        return load_dpv(p)[1]
    if p.endswith(".txt"):
        m = load_matrix(p)
        return m if nvertices is None or m.shape[1] == nvertices else m.T
    return load_metric(p, nvertices)


if __name__ == "__main__":  # python -m newmsm_amd.meshio file.gii: list the arrays
    for intent, a in read_gifti(sys.argv[1]):
        print(intent, a.dtype, a.shape)
