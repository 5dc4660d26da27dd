// msmhip_io.hpp -- the mesh / metric files on either side of the path, for a C++ host (header only, C++17; link -lz -lexpat):
//
//   GIFTI (.surf.gii, .func.gii, .shape.gii) as Mesh::load_gifti / Mesh::save_gifti read and write it
//   (/root/reference/libraries/msm-newresampler/src/mesh.cpp:350-398, 582-631): a surface is a NIFTI_INTENT_POINTSET array
//   (float32, N x 3) followed by a NIFTI_INTENT_TRIANGLE array (int32, T x 3); a metric file holds one float32 array of N
//   values per feature; files are written GZipBase64Binary, row-major, little-endian (:607-609: surfaces go out as float32).
//   The reader also accepts ASCII and Base64Binary encodings, big-endian data, column-major arrays and the other numeric
//   GIFTI datatypes.
//   FreeSurfer ASCII (.asc) as Mesh::load_ascii reads it (mesh.cpp:455-515): "#!ascii" header, "NVertices NFaces", then
//   "x y z value" per vertex and "a b c value" per face.
//
// newMSM reads GIFTI through FSL's giftiInterface, which is not part of the reference tree: this follows the GIFTI 1.0
// specification and the reference's call sites.  newmsm_amd/meshio.py is the same in Python; tests/test_cpp_io.py checks that
// the two writers produce the same bytes and that each reads what the other wrote.  Containers are those of msmhip.hpp:
// Points AoS (x0 y0 z0 ...), Triangles AoS, Matrix row-major D x V (newresampler::Mesh::pvalues).
#ifndef MSMHIP_IO_HPP
#define MSMHIP_IO_HPP

#include <expat.h>
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace msmhip {
namespace io {

struct Error : std::runtime_error {  // MeshException (what() carries the message)
    using std::runtime_error::runtime_error;
};

struct DataArray {
    std::string intent;        // NIFTI_INTENT_*
    std::vector<int64_t> dims;
    bool integral = false;     // the file's DataType is an integer type
    std::vector<double> values;  // row-major, converted to double (float32 values exactly)
};

namespace detail {

inline std::string b64_encode(const unsigned char *p, size_t n) {
    static const char tab[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    std::string out;
    out.reserve((n + 2) / 3 * 4);
    for (size_t i = 0; i < n; i += 3) {
        const unsigned v = (unsigned)p[i] << 16 | (i + 1 < n ? (unsigned)p[i + 1] << 8 : 0u) | (i + 2 < n ? (unsigned)p[i + 2] : 0u);
        out.push_back(tab[v >> 18 & 63]);
        out.push_back(tab[v >> 12 & 63]);
        out.push_back(i + 1 < n ? tab[v >> 6 & 63] : '=');
        out.push_back(i + 2 < n ? tab[v & 63] : '=');
    }
    return out;
}
inline std::vector<unsigned char> b64_decode(const std::string &s) {
    std::vector<unsigned char> out;
    out.reserve(s.size() * 3 / 4);
    unsigned acc = 0;
    int bits = 0;
    for (const char ch : s) {
        int v;
        if (ch >= 'A' && ch <= 'Z') v = ch - 'A';
        else if (ch >= 'a' && ch <= 'z') v = ch - 'a' + 26;
        else if (ch >= '0' && ch <= '9') v = ch - '0' + 52;
        else if (ch == '+') v = 62;
        else if (ch == '/') v = 63;
        else if (ch == '=') break;
        else continue;  // white space
        acc = acc << 6 | (unsigned)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out.push_back((unsigned char)(acc >> bits & 0xff));
        }
    }
    return out;
}
inline std::vector<unsigned char> inflate_all(const std::vector<unsigned char> &in, size_t expect) {
    std::vector<unsigned char> out(expect ? expect : 1);
    for (;;) {
        uLongf n = (uLongf)out.size();
        const int rc = uncompress(out.data(), &n, in.data(), (uLong)in.size());
        if (rc == Z_OK) {
            out.resize(n);
            return out;
        }
        if (rc != Z_BUF_ERROR) throw Error("GIFTI: the compressed data block is corrupt");
        out.resize(out.size() * 2 + 64);
    }
}

struct TypeInfo {
    const char *name;
    int size;
    char kind;  // 'f' float, 'i' signed, 'u' unsigned
};
inline const TypeInfo *type_of(const std::string &name) {
    static const TypeInfo types[] = {{"NIFTI_TYPE_UINT8", 1, 'u'},   {"NIFTI_TYPE_INT8", 1, 'i'},   {"NIFTI_TYPE_INT16", 2, 'i'}, {"NIFTI_TYPE_UINT16", 2, 'u'},
                                     {"NIFTI_TYPE_INT32", 4, 'i'},   {"NIFTI_TYPE_UINT32", 4, 'u'}, {"NIFTI_TYPE_INT64", 8, 'i'}, {"NIFTI_TYPE_UINT64", 8, 'u'},
                                     {"NIFTI_TYPE_FLOAT32", 4, 'f'}, {"NIFTI_TYPE_FLOAT64", 8, 'f'}};
    for (const auto &t : types)
        if (name == t.name) return &t;
    return nullptr;
}
inline double decode_value(const unsigned char *p, const TypeInfo &t, bool big_endian) {
    unsigned char b[8];
    for (int k = 0; k < t.size; ++k) b[k] = big_endian ? p[t.size - 1 - k] : p[k];  // to little endian (the host's order on every target here)
    if (t.kind == 'f') {
        if (t.size == 4) {
            float f;
            std::memcpy(&f, b, 4);
            return (double)f;
        }
        double d;
        std::memcpy(&d, b, 8);
        return d;
    }
    uint64_t u = 0;
    for (int k = t.size - 1; k >= 0; --k) u = u << 8 | b[k];
    if (t.kind == 'u') return (double)u;
    const int shift = 64 - 8 * t.size;
    return (double)((int64_t)(u << shift) >> shift);
}

struct Parser {
    std::vector<DataArray> arrays;
    // the array being read
    bool in_array = false, in_data = false;
    std::string intent, dtype, encoding, endian, order, text;
    std::vector<int64_t> dims;
    std::string root;
    std::string error;

    static void start(void *ud, const XML_Char *name, const XML_Char **atts) {
        Parser &p = *static_cast<Parser *>(ud);
        if (p.root.empty()) p.root = name;
        if (std::strcmp(name, "DataArray") == 0) {
            p.in_array = true;
            p.intent = "NIFTI_INTENT_NONE", p.dtype.clear(), p.encoding = "ASCII", p.endian = "LittleEndian", p.order = "RowMajorOrder";
            int ndim = 1;
            std::vector<std::pair<int, int64_t>> d;
            for (int i = 0; atts[i]; i += 2) {
                const std::string k = atts[i], v = atts[i + 1];
                if (k == "Intent") p.intent = v;
                else if (k == "DataType") p.dtype = v;
                else if (k == "Encoding") p.encoding = v;
                else if (k == "Endian") p.endian = v;
                else if (k == "ArrayIndexingOrder") p.order = v;
                else if (k == "Dimensionality") ndim = std::atoi(v.c_str());
                else if (k.size() > 3 && k.compare(0, 3, "Dim") == 0 && k[3] >= '0' && k[3] <= '9') d.emplace_back(std::atoi(k.c_str() + 3), std::atoll(v.c_str()));
            }
            p.dims.assign((size_t)ndim, 0);
            for (const auto &e : d)
                if (e.first >= 0 && e.first < ndim) p.dims[(size_t)e.first] = e.second;
        } else if (p.in_array && std::strcmp(name, "Data") == 0) {
            p.in_data = true;
            p.text.clear();
        }
    }
    static void chars(void *ud, const XML_Char *s, int len) {
        Parser &p = *static_cast<Parser *>(ud);
        if (p.in_data) p.text.append(s, (size_t)len);
    }
    static void end(void *ud, const XML_Char *name) {
        Parser &p = *static_cast<Parser *>(ud);
        if (std::strcmp(name, "Data") == 0) p.in_data = false;
        if (std::strcmp(name, "DataArray") == 0 && p.in_array) {
            p.in_array = false;
            if (p.error.empty()) p.finish();
        }
    }
    void finish() {
        const TypeInfo *t = type_of(dtype);
        if (!t) {
            error = "GIFTI: unsupported DataType '" + dtype + "'";
            return;
        }
        size_t count = dims.empty() ? 0 : 1;
        for (const int64_t d : dims) count *= (size_t)(d > 0 ? d : 0);
        DataArray a;
        a.intent = intent;
        a.dims = dims;
        a.integral = t->kind != 'f';
        std::vector<double> flat;
        if (encoding == "ASCII") {
            std::istringstream in(text);
            double v;
            while (in >> v) flat.push_back(t->kind == 'f' && t->size == 4 ? (double)(float)v : v);
        } else if (encoding == "Base64Binary" || encoding == "GZipBase64Binary") {
            std::vector<unsigned char> raw = b64_decode(text);
            if (encoding == "GZipBase64Binary") raw = inflate_all(raw, count * (size_t)t->size);
            const size_t n = raw.size() / (size_t)t->size;
            flat.resize(n);
            const bool big = endian != "LittleEndian";
            for (size_t i = 0; i < n; ++i) flat[i] = decode_value(raw.data() + i * (size_t)t->size, *t, big);
        } else {
            error = "GIFTI: Encoding '" + encoding + "' is not supported (external files are not read)";
            return;
        }
        if (flat.size() != count) {
            error = "GIFTI: array holds " + std::to_string(flat.size()) + " values, its dimensions say " + std::to_string(count);
            return;
        }
        if (order == "ColumnMajorOrder" && dims.size() == 2) {
            a.values.resize(count);
            const size_t R = (size_t)dims[0], C = (size_t)dims[1];
            for (size_t r = 0; r < R; ++r)
                for (size_t c = 0; c < C; ++c) a.values[r * C + c] = flat[c * R + r];
        } else {
            a.values = std::move(flat);
        }
        arrays.push_back(std::move(a));
    }
};

inline std::string slurp(const std::string &path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw Error("cannot open " + path);
    std::ostringstream ss;
    ss << in.rdbuf();
    return ss.str();
}
inline bool ends_with(const std::string &s, const char *suffix) {
    const size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

inline std::string encode_array(const char *intent, const char *dtype, const void *data, size_t bytes, const std::vector<int64_t> &dims, bool with_coordsys) {
    uLongf zn = compressBound((uLong)bytes);
    std::vector<unsigned char> z(zn);
    if (compress(z.data(), &zn, static_cast<const Bytef *>(data), (uLong)bytes) != Z_OK) throw Error("GIFTI: compression failed");
    std::ostringstream o;
    o << "   <DataArray Intent=\"" << intent << "\" DataType=\"" << dtype << "\" ArrayIndexingOrder=\"RowMajorOrder\" Dimensionality=\"" << dims.size() << "\"";
    for (size_t k = 0; k < dims.size(); ++k) o << " Dim" << k << "=\"" << dims[k] << "\"";
    o << " Encoding=\"GZipBase64Binary\" Endian=\"LittleEndian\" ExternalFileName=\"\" ExternalFileOffset=\"\">\n      <MetaData/>\n";
    if (with_coordsys)
        o << "      <CoordinateSystemTransformMatrix>\n         <DataSpace><![CDATA[NIFTI_XFORM_UNKNOWN]]></DataSpace>\n"
             "         <TransformedSpace><![CDATA[NIFTI_XFORM_UNKNOWN]]></TransformedSpace>\n"
             "         <MatrixData>1.000000 0.000000 0.000000 0.000000 0.000000 1.000000 0.000000 0.000000 0.000000 0.000000 1.000000 0.000000 0.000000 0.000000 "
             "0.000000 1.000000</MatrixData>\n      </CoordinateSystemTransformMatrix>\n";
    o << "      <Data>" << b64_encode(z.data(), zn) << "</Data>\n   </DataArray>";
    return o.str();
}
inline void write_gifti(const std::string &path, const std::vector<std::string> &arrays) {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw Error("cannot write " + path);
    f << "<?xml version=\"1.0\" encoding=\"UTF-8\"?>\n<!DOCTYPE GIFTI SYSTEM \"http://www.nitrc.org/frs/download.php/115/gifti.dtd\">\n"
      << "<GIFTI Version=\"1.0\" NumberOfDataArrays=\"" << arrays.size() << "\">\n   <MetaData/>\n   <LabelTable/>\n";
    for (const auto &a : arrays) f << a << "\n";
    f << "</GIFTI>\n";
}

struct Ascii {
    std::vector<double> xyz, values;
    std::vector<int32_t> tri;
};
inline Ascii read_ascii(const std::string &path) {  // Mesh::load_ascii, R/mesh.cpp:455-515
    std::ifstream in(path);
    if (!in) throw Error("cannot open " + path);
    std::string header;
    std::getline(in, header);
    if (header.find("#!ascii") == std::string::npos) throw Error("Mesh::load_ascii:error in the header");
    long nv = -1, nf = -1;
    in >> nv >> nf;
    if (!in || nv < 0 || nf < 0) throw Error("Mesh::load_ascii: " + path + " is truncated");
    Ascii a;
    a.xyz.resize(3 * (size_t)nv), a.values.resize((size_t)nv), a.tri.resize(3 * (size_t)nf);
    for (long i = 0; i < nv; ++i) {
        double v;
        in >> a.xyz[3 * (size_t)i] >> a.xyz[3 * (size_t)i + 1] >> a.xyz[3 * (size_t)i + 2] >> v;
        a.values[(size_t)i] = (double)(float)v;  // the value passes through a float (:486)
    }
    for (long i = 0; i < nf; ++i) {
        double p0, p1, p2, v;
        in >> p0 >> p1 >> p2 >> v;
        a.tri[3 * (size_t)i] = (int32_t)p0, a.tri[3 * (size_t)i + 1] = (int32_t)p1, a.tri[3 * (size_t)i + 2] = (int32_t)p2;
    }
    if (!in) throw Error("Mesh::load_ascii: " + path + " is truncated");
    return a;
}

}  // namespace detail

// all data arrays of a GIFTI file, in file order
inline std::vector<DataArray> read_gifti(const std::string &path) {
    const std::string xml = detail::slurp(path);
    detail::Parser p;
    XML_Parser xp = XML_ParserCreate(nullptr);
    XML_SetUserData(xp, &p);
    XML_SetElementHandler(xp, detail::Parser::start, detail::Parser::end);
    XML_SetCharacterDataHandler(xp, detail::Parser::chars);
    const bool ok = XML_Parse(xp, xml.data(), (int)xml.size(), 1) != XML_STATUS_ERROR;
    const std::string xml_err = ok ? "" : XML_ErrorString(XML_GetErrorCode(xp));
    XML_ParserFree(xp);
    if (!ok) throw Error("GIFTI: " + path + " is not well-formed XML (" + xml_err + ")");
    if (p.root != "GIFTI") throw Error("GIFTI: " + path + " has root element <" + p.root + ">");
    if (!p.error.empty()) throw Error(p.error);
    return std::move(p.arrays);
}

// (xyz AoS 3 x V, tri AoS 3 x T) of a .surf.gii or FreeSurfer .asc file
inline std::pair<std::vector<double>, std::vector<int32_t>> load_surface(const std::string &path) {
    if (detail::ends_with(path, ".asc")) {
        detail::Ascii a = detail::read_ascii(path);
        return {std::move(a.xyz), std::move(a.tri)};
    }
    const DataArray *pts = nullptr, *tris = nullptr;
    const std::vector<DataArray> arrays = read_gifti(path);
    for (const auto &a : arrays) {
        if (!pts && a.intent == "NIFTI_INTENT_POINTSET") pts = &a;
        if (!tris && a.intent == "NIFTI_INTENT_TRIANGLE") tris = &a;
    }
    if (!pts || !tris) throw Error("GIFTI: " + path + " holds no surface (POINTSET + TRIANGLE arrays)");
    if (pts->dims.size() != 2 || pts->dims[1] != 3 || tris->dims.size() != 2 || tris->dims[1] != 3) throw Error("GIFTI: surface arrays must be N x 3");
    std::vector<int32_t> tri(tris->values.size());
    const double V = (double)pts->dims[0];
    for (size_t i = 0; i < tri.size(); ++i) {
        if (tris->values[i] < 0 || tris->values[i] >= V) throw Error("GIFTI: triangle refers to a vertex that does not exist");
        tri[i] = (int32_t)tris->values[i];
    }
    return {pts->values, tri};
}

// save_gifti for a '.surf' file: float32 coordinates (as the reference writes them), int32 triangles; .asc: FreeSurfer ASCII
inline void save_ascii(const std::string &path, const std::vector<double> &xyz, const std::vector<int32_t> &tri, const std::vector<double> *values = nullptr) {
    FILE *f = std::fopen(path.c_str(), "w");
    if (!f) throw Error("cannot write " + path);
    const size_t V = xyz.size() / 3, T = tri.size() / 3;
    std::fprintf(f, "#!ascii from msm-mi355x\n%zu %zu\n", V, T);
    for (size_t i = 0; i < V; ++i) std::fprintf(f, "%.17g %.17g %.17g %.9g\n", xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], values ? (*values)[i] : 0.0);
    for (size_t i = 0; i < T; ++i) std::fprintf(f, "%d %d %d 0\n", tri[3 * i], tri[3 * i + 1], tri[3 * i + 2]);
    std::fclose(f);
}
inline void save_surface(const std::string &path, const std::vector<double> &xyz, const std::vector<int32_t> &tri) {
    if (detail::ends_with(path, ".asc")) return save_ascii(path, xyz, tri);
    std::vector<float> p(xyz.begin(), xyz.end());
    const std::vector<int64_t> dp{(int64_t)(xyz.size() / 3), 3}, dt{(int64_t)(tri.size() / 3), 3};
    detail::write_gifti(path, {detail::encode_array("NIFTI_INTENT_POINTSET", "NIFTI_TYPE_FLOAT32", p.data(), p.size() * 4, dp, true),
                               detail::encode_array("NIFTI_INTENT_TRIANGLE", "NIFTI_TYPE_INT32", tri.data(), tri.size() * 4, dt, true)});
}

// D x V row-major matrix of a .func.gii / .shape.gii (one array per feature; an N x K array counts as K features) or of the value
// column of an .asc file.  *D receives the number of rows.  nvertices >= 0: checked against the arrays (R/mesh.cpp:392).
inline std::vector<double> load_metric(const std::string &path, int *D, long nvertices = -1) {
    if (detail::ends_with(path, ".asc")) {
        detail::Ascii a = detail::read_ascii(path);
        if (D) *D = 1;
        return std::move(a.values);
    }
    std::vector<std::vector<double>> rows;
    for (const auto &a : read_gifti(path)) {
        if (a.intent == "NIFTI_INTENT_POINTSET" || a.intent == "NIFTI_INTENT_TRIANGLE") continue;
        const size_t n = a.dims.empty() ? 0 : (size_t)a.dims[0], K = n ? a.values.size() / n : 0;
        if (nvertices >= 0 && (long)n != nvertices) throw Error(" mismatch between data and surface dimensions");
        for (size_t k = 0; k < K; ++k) {
            std::vector<double> r(n);
            for (size_t i = 0; i < n; ++i) r[i] = a.values[i * K + k];
            rows.push_back(std::move(r));
        }
    }
    if (rows.empty()) throw Error("GIFTI: " + path + " holds no data arrays");
    for (const auto &r : rows)
        if (r.size() != rows[0].size()) throw Error(" mismatch between data and surface dimensions");
    std::vector<double> out;
    out.reserve(rows.size() * rows[0].size());
    for (const auto &r : rows) out.insert(out.end(), r.begin(), r.end());
    if (D) *D = (int)rows.size();
    return out;
}

// save_gifti for a '.func' / '.shape' file: one float32 NIFTI_INTENT_NONE array per feature row
inline void save_metric(const std::string &path, const std::vector<double> &data, int D) {
    const size_t V = D > 0 ? data.size() / (size_t)D : 0;
    std::vector<std::string> arrays;
    for (int d = 0; d < D; ++d) {
        std::vector<float> row(data.begin() + (size_t)d * V, data.begin() + (size_t)(d + 1) * V);
        arrays.push_back(detail::encode_array("NIFTI_INTENT_NONE", "NIFTI_TYPE_FLOAT32", row.data(), row.size() * 4, {(int64_t)V}, false));
    }
    detail::write_gifti(path, arrays);
}

// ---- the text formats of -f ASCII / ASCII_MAT (set_output_format, M/mesh_registration.cpp:827-842)
namespace detail {
// a float as std::ostream writes it by default: the value rounded to float, six significant digits (%g)
inline std::string g(double x) {
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%g", (double)(float)x);
    return buf;
}
// the numbers of a text file, row by row (lines starting with '#' and empty lines skipped)
inline std::vector<std::vector<double>> read_rows(const std::string &path) {
    std::ifstream in(path);
    if (!in) throw Error("cannot open " + path);
    std::vector<std::vector<double>> rows;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::vector<double> r;
        std::string tok;
        while (ls >> tok) {
            if (tok[0] == '#') break;
            char *end = nullptr;
            const double v = std::strtod(tok.c_str(), &end);
            if (end == tok.c_str()) throw Error(path + ": not a number: " + tok);
            r.push_back(v);
        }
        if (!r.empty()) rows.push_back(std::move(r));
    }
    return rows;
}
}  // namespace detail

// Mesh::save_dpv, R/mesh.cpp:707-741: `index x y z value` per vertex (indices below 100 zero-padded to three digits), first data row only
inline void save_dpv(const std::string &path, const std::vector<double> &xyz, const std::vector<double> &data) {
    const size_t V = xyz.size() / 3;
    if (data.size() < V) throw Error("Mesh::save_dpv, data and mesh dimensions do not agree");
    FILE *f = std::fopen(path.c_str(), "w");
    if (!f) throw Error("cannot write " + path);
    for (size_t i = 0; i < V; ++i) {
        if (i < 100) std::fprintf(f, "%03zu", i);
        else std::fprintf(f, "%zu", i);
        std::fprintf(f, " %s %s %s %s\n", detail::g(xyz[3 * i]).c_str(), detail::g(xyz[3 * i + 1]).c_str(), detail::g(xyz[3 * i + 2]).c_str(), detail::g(data[i]).c_str());
    }
    std::fclose(f);
}
// Mesh::save_matrix, R/mesh.cpp:743-766: one line per data row, values separated (and followed) by a blank
inline void save_matrix(const std::string &path, const std::vector<double> &data, int D) {
    const size_t V = D > 0 ? data.size() / (size_t)D : 0;
    FILE *f = std::fopen(path.c_str(), "w");
    if (!f) throw Error("cannot write " + path);
    for (int d = 0; d < D; ++d) {
        for (size_t i = 0; i < V; ++i) std::fprintf(f, "%s ", detail::g(data[(size_t)d * V + i]).c_str());
        std::fprintf(f, "\n");
    }
    std::fclose(f);
}
// the data of a --indata / --refdata file by its extension, D x V row-major (set_data, M/reg_tools.cpp:846-867 / Mesh::load, R/mesh.cpp:296-348):
// .dpv: the value column of `index x y z value` lines; .txt: a matrix, one row per feature (or per vertex: transposed when the rows are as long as
// nvertices says they should not be); anything else: load_metric
inline std::vector<double> load_data(const std::string &path, int *D, long nvertices = -1) {
    if (detail::ends_with(path, ".dpv")) {
        const auto rows = detail::read_rows(path);
        std::vector<double> out;
        for (const auto &r : rows) {
            if (r.size() != 5) throw Error("Mesh::load_dpv:error opening file (wrong format) : " + path);
            out.push_back(r[4]);
        }
        if (D) *D = 1;
        return out;
    }
    if (detail::ends_with(path, ".txt")) {
        const auto rows = detail::read_rows(path);
        if (rows.empty()) throw Error(path + " holds no data");
        const size_t R = rows.size(), C = rows[0].size();
        for (const auto &r : rows)
            if (r.size() != C) throw Error(path + ": rows of different lengths");
        const bool transpose = nvertices >= 0 && (long)C != nvertices;
        std::vector<double> out(R * C);
        for (size_t i = 0; i < R; ++i)
            for (size_t j = 0; j < C; ++j) out[transpose ? j * R + i : i * C + j] = rows[i][j];
        if (D) *D = (int)(transpose ? C : R);
        return out;
    }
    return load_metric(path, D, nvertices);
}

}  // namespace io
}  // namespace msmhip

#endif
