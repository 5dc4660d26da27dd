// newmsm.cpp -- the `newmsm` executable over the MI355X path, host side in C++ (north_star: "host code stays C++ and calls HIP through a thin C-ABI ...
// keeping the `newmsm` CLI, config-file format and sphere.reg / warp output layout").  What CLI/newmsm.cpp:6-59 does with Mesh_registration /
// Group_Mesh_registration, assembled from this repository's headers:
//
//   flags            src/msmOptions.h:59-157 (short and long forms, `--key=value` or `--key value`); -h / --help prints them
//   pairwise         set_input / set_reference (load, recentre, true_rescale to RAD: M/mesh_registration.cpp:416-438), set_anatomical as loaded (:434-438),
//                    the configuration through the reference's grammar (msmhip_config.hpp = parse_reg_options :459-784), run_multiresolutions
//                    (msmhip_registration.hpp = :30-50), then <out>sphere.reg<surf>, <out>sphere.LR.reg<surf>, <out>transformed_and_reprojected<data>
//                    (:47-49, :352-408, M/mesh_registration.h:170)
//   -g / --groupwise --meshes / --data path lists (read_ascii_list :871-884), --template, --mask; per subject <out>sphere-<i>.reg<surf>,
//                    <out>sphere-<i>.LR.reg<surf>, <out>transformed_and_reprojected-<i><data> (M/group_mesh_registration.cpp:120-133, .h:79-82)
//   -f               GIFTI (.surf.gii / .func.gii), ASCII (.asc / .dpv), ASCII_MAT (.asc / .txt) as set_output_format names them (:827-842)
//
// Outside the path and said so instead of silently dropped: AFFINE / RIGID levels (skipped with a note on stderr), --trans, VTK output; the binary solve of
// --dopt=HOCR / FastPD is a stand-in (iterated conditional modes: FastPD and ELC are licence-restricted and FSL-bound), so a run exercises the path exactly as
// newmsm would but its labelings are not HOCR's.  tools/register_files.py is the same program in Python; tests/test_gpu_registration.py compares their files.
//
// Errors: a MeshregException's message on stderr, exit status 1 (CLI/newmsm.cpp:62-68).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "msmhip_config.hpp"
#include "msmhip_group_registration.hpp"
#include "msmhip_io.hpp"

using namespace msmhip;

namespace {

constexpr double RAD = 100.0;

struct Flag {
    const char *shortname, *longname, *help;
    bool takes_value;
};
// src/msmOptions.h:59-157, in the reference's order
const Flag kFlags[] = {
    {"-h", "--help", "display this message", false},
    {"-v", "--verbose", "switch on diagnostic messages", false},
    {"-p", "--printoptions", "print configuration file options", false},
    {"-d", "--debug", "run debugging or optimising options", false},
    {"-g", "--groupwise", "run newMSM in groupwise mode", false},
    {"-m", "--meshes", "groupwise mode only; list of paths to input meshes. Needs to be a sphere", true},
    {"-s", "--template", "groupwise mode only; templates sphere for resampling. Needs to be a sphere", true},
    {"-l", "--data", "groupwise mode only; list of paths of the data files", true},
    {"-k", "--mask", "groupwise mode only; mask file path", true},
    {"-M", "--inmesh", "input mesh (available formats: ASCII, GIFTI). Needs to be a sphere", true},
    {"-R", "--refmesh", "reference mesh (available formats: ASCII, GIFTI). Needs to be a sphere. If not included algorithm assumes reference mesh is equivalent input", true},
    {"-a", "--inanat", "input anatomical mesh (must either supply both input and reference anatomical surfaces or none)", true},
    {"-A", "--refanat", "reference anatomical mesh", true},
    {"-i", "--indata", "scalar or multivariate data for input - can be ASCII (.asc,.dpv,.txt) or GIFTI (.func.gii or .shape.gii)", true},
    {"-I", "--refdata", "scalar or multivariate data for reference", true},
    {"-t", "--trans", "Transformed source mesh (output of a previous registration): not supported here", true},
    {"-w", "--inweight", "cost function weighting for input", true},
    {"-W", "--refweight", "cost function weighting for reference", true},
    {"-o", "--out", "output basename", true},
    {"-f", "--format", "format of output files, can be: GIFTI, ASCII or ASCII_MAT", true},
    {"-c", "--conf", "configuration file", true},
    {nullptr, "--device", "the GPU to run on (default 0)", true},
};

void usage() {
    std::cout << "\nnewmsm [options]   (msm-mi355x: newMSM's DISCRETE path on an MI355X)\n\n";
    for (const Flag &f : kFlags) std::cout << "\t" << (f.shortname ? std::string(f.shortname) + "," : std::string()) << f.longname << "\t" << f.help << "\n";
    std::cout << std::endl;
}

struct Options {
    std::map<std::string, std::string> value;  // by long name without the dashes
    bool has(const std::string &k) const { return value.count(k) != 0; }
    std::string get(const std::string &k) const {
        auto it = value.find(k);
        return it == value.end() ? std::string() : it->second;
    }
};

Options parse_command_line(int argc, char **argv) {
    Options o;
    for (int i = 1; i < argc; ++i) {
        std::string arg = argv[i], val;
        bool inline_value = false;
        const size_t eq = arg.find('=');
        if (arg.rfind("--", 0) == 0 && eq != std::string::npos) {
            val = arg.substr(eq + 1);
            arg = arg.substr(0, eq);
            inline_value = true;
        }
        const Flag *hit = nullptr;
        for (const Flag &f : kFlags)
            if (arg == f.longname || (f.shortname && arg == f.shortname)) hit = &f;
        if (!hit) throw Error(MSM_ERR_INVALID, "Option " + arg + " is not an option");  // X_OptionError
        const std::string key = std::string(hit->longname).substr(2);
        if (!hit->takes_value) {
            o.value[key] = "1";
            continue;
        }
        if (!inline_value) {
            if (i + 1 >= argc) throw Error(MSM_ERR_INVALID, "Option " + arg + " requires an argument");
            val = argv[++i];
        }
        o.value[key] = val;
    }
    return o;
}

// recentre + true_rescale (R/mesh.cpp:1198-1255), as Mesh_registration::set_input / set_reference apply them: the mean of the vertices to the origin
// (summed vertex by vertex), every vertex to the radius
Points on_sphere(Points xyz, double rad = RAD) {
    const size_t V = xyz.size() / 3;
    double c[3] = {0.0, 0.0, 0.0};
    for (size_t i = 0; i < V; ++i)
        for (int a = 0; a < 3; ++a) c[a] += xyz[3 * i + a];
    for (int a = 0; a < 3; ++a) c[a] /= (double)V;
    for (size_t i = 0; i < V; ++i) {
        double p[3];
        for (int a = 0; a < 3; ++a) p[a] = xyz[3 * i + a] - c[a];
        const double n = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]), k = rad / n;
        for (int a = 0; a < 3; ++a) xyz[3 * i + a] = p[a] * k;
    }
    return xyz;
}

std::string slurp(const std::string &path) {
    if (path.empty()) return std::string();
    std::ifstream in(path);
    if (!in) throw Error(MSM_ERR_INVALID, "cannot open " + path);
    std::stringstream ss;
    ss << in.rdbuf();
    return ss.str();
}

// Mesh_registration::read_ascii_list, M/mesh_registration.cpp:871-884: the whitespace-separated entries of a text file
std::vector<std::string> read_ascii_list(const std::string &path) {
    std::istringstream in(slurp(path));
    std::vector<std::string> out;
    std::string tok;
    while (in >> tok) out.push_back(tok);
    return out;
}

struct Formats {
    std::string surf, data;
};
Formats output_formats(const std::string &fmt) {  // set_output_format, M/mesh_registration.cpp:827-842
    if (fmt == "GIFTI") return {".surf.gii", ".func.gii"};
    if (fmt == "ASCII") return {".asc", ".dpv"};
    if (fmt == "ASCII_MAT") return {".asc", ".txt"};
    throw Error(MSM_ERR_INVALID, "newmsm: VTK output is not written here (GIFTI, ASCII, ASCII_MAT)");
}

void save_data(const std::string &path, const Points &mesh_xyz, const Matrix &data, int D) {
    const size_t n = path.size();
    if (n >= 4 && path.compare(n - 4, 4, ".dpv") == 0) io::save_dpv(path, mesh_xyz, data);
    else if (n >= 4 && path.compare(n - 4, 4, ".txt") == 0) io::save_matrix(path, data, D);
    else io::save_metric(path, data, D);
}

void note_skipped(const std::vector<std::pair<int, std::string>> &skipped) {
    for (const auto &sk : skipped)
        std::cerr << "newmsm: level " << sk.first + 1 << " (--opt=" << sk.second << ") is outside the path (the affine stage stays on the CPU in newmsm): skipped" << std::endl;
}

int run_pairwise(const Options &o, const Formats &fmt, int device) {
    for (const char *flag : {"inmesh", "indata", "refdata"})
        if (o.get(flag).empty()) throw Error(MSM_ERR_INVALID, std::string("newmsm: --") + flag + " is required");
    if (o.has("trans")) throw Error(MSM_ERR_INVALID, "newmsm: --trans (a previous registration as the starting point) is not wired into the level loop");
    if (o.get("inanat").empty() != o.get("refanat").empty()) throw Error(MSM_ERR_INVALID, "Error: must supply both anatomical meshes or none");  // CLI/newmsm.cpp:41-43
    auto [ixyz0, itri] = io::load_surface(o.get("inmesh"));
    auto [rxyz0, rtri] = io::load_surface(o.get("refmesh").empty() ? o.get("inmesh") : o.get("refmesh"));
    const Points ixyz = on_sphere(ixyz0), rxyz = on_sphere(rxyz0);
    int D = 0, Dr = 0;
    const Matrix idata = io::load_data(o.get("indata"), &D, (long)(ixyz.size() / 3)), rdata = io::load_data(o.get("refdata"), &Dr, (long)(rxyz.size() / 3));
    if (D != Dr) throw Error(MSM_ERR_INVALID, "Mesh_registration: input and reference data have different numbers of feature rows (" + std::to_string(D) + ", " + std::to_string(Dr) + ")");
    const bool anat = !o.get("inanat").empty();
    bool varnorm = false;
    std::vector<std::pair<int, std::string>> skipped;
    const std::vector<LevelSpec> levels = levels_from_config(parse_config(slurp(o.get("conf")), o.get("conf").empty()), D, &varnorm, &skipped, anat);
    note_skipped(skipped);
    if (levels.empty()) throw Error(MSM_ERR_INVALID, "newmsm: the configuration holds no DISCRETE level");
    Points in_anat, ref_anat;
    if (anat) {  // set_anatomical: loaded as they are
        in_anat = io::load_surface(o.get("inanat")).first;
        ref_anat = io::load_surface(o.get("refanat")).first;
    }
    Matrix in_w, ref_w;
    int in_wr = 0, ref_wr = 0;
    const bool weighted = !o.get("inweight").empty() && !o.get("refweight").empty();
    if (weighted) {
        in_w = io::load_data(o.get("inweight"), &in_wr, (long)(ixyz.size() / 3));
        ref_w = io::load_data(o.get("refweight"), &ref_wr, (long)(rxyz.size() / 3));
    }
    Context ctx(device);
    if (o.has("verbose"))
        std::cout << "This is newMSM's DISCRETE path on an MI355X (msm-mi355x).\nStarting multiresolution with " << levels.size() << " levels." << std::endl;
    const MultiresResult res = run_multiresolutions(ctx, ixyz, itri, idata, rxyz, rtri, rdata, D, levels, varnorm, nullptr, anat ? &in_anat : nullptr,
                                                    anat ? &ref_anat : nullptr, weighted ? &in_w : nullptr, in_wr, weighted ? &ref_w : nullptr, ref_wr);
    const std::string out = o.get("out");
    io::save_surface(out + "sphere.reg" + fmt.surf, res.sphere_reg, itri);  // transform
    const Triangles last_tri = make_mesh_from_icosa(levels.back().data_order).second;
    io::save_surface(out + "sphere.LR.reg" + fmt.surf, res.level_reg.back(), last_tri);  // saveSPH_reg
    Mesh moved(ctx, res.sphere_reg, itri), target(ctx, rxyz, rtri);
    save_data(out + "transformed_and_reprojected" + fmt.data, rxyz, metric_resample(moved, idata, target), D);  // save_transformed_data
    if (o.has("verbose"))
        for (size_t k = 0; k < res.energies.size(); ++k) {
            std::cout << "level " << k + 1 << ": energies per iteration";
            for (double e : res.energies[k]) std::printf(" %.4f", e);
            std::cout << std::endl;
        }
    return 0;
}

int run_groupwise(const Options &o, const Formats &fmt, int device) {
    for (const char *flag : {"meshes", "template", "data"})
        if (o.get(flag).empty()) throw Error(MSM_ERR_INVALID, std::string("newmsm: --groupwise needs --") + flag);
    const std::vector<std::string> mesh_files = read_ascii_list(o.get("meshes")), data_files = read_ascii_list(o.get("data"));
    if (mesh_files.size() != data_files.size())
        throw Error(MSM_ERR_INVALID, "featurespace::Initialize do not have the same number of datasets and surface meshes");  // M/featurespace.cpp:43-44
    const Config cfg = parse_config(slurp(o.get("conf")), o.get("conf").empty());
    bool varnorm = false;
    const std::vector<GroupLevelSpec> levels = group_levels_from_config(cfg, &varnorm);  // refuses AFFINE / RIGID levels and optimisers other than HOCR
    if (levels.empty()) throw Error(MSM_ERR_INVALID, "newmsm: the configuration holds no DISCRETE level");
    std::vector<std::pair<Points, Triangles>> meshes;
    for (size_t k = 0; k < mesh_files.size(); ++k) {
        if (o.has("verbose")) std::cout << "Mesh #" << k << " is " << mesh_files[k] << std::endl;
        auto [xyz, tri] = io::load_surface(mesh_files[k]);
        meshes.emplace_back(on_sphere(xyz), tri);
    }
    if (o.has("verbose")) std::cout << "Template is " << o.get("template") << std::endl;
    auto [txyz0, ttri] = io::load_surface(o.get("template"));
    const Points txyz = on_sphere(txyz0);
    std::vector<Matrix> datas;
    int D = 0;
    for (size_t k = 0; k < data_files.size(); ++k) {
        int Dk = 0;
        datas.push_back(io::load_data(data_files[k], &Dk, (long)(meshes[k].first.size() / 3)));
        if (k == 0) D = Dk;
        else if (Dk != D) throw Error(MSM_ERR_INVALID, "newmsm: the subjects' data have different numbers of feature rows");
    }
    std::vector<double> mask;
    if (!o.get("mask").empty()) {
        int Dm = 0;
        const Matrix m = io::load_data(o.get("mask"), &Dm, (long)(txyz.size() / 3));
        mask.assign(m.begin(), m.begin() + (long)(txyz.size() / 3));  // the first row
    }
    Context ctx(device);
    if (o.has("verbose"))
        std::cout << "This is newMSM's groupwise DISCRETE path on an MI355X (msm-mi355x).\nStarting multiresolution with " << levels.size() << " levels." << std::endl;
    const GroupMultiresResult res = run_group_multiresolutions(ctx, meshes, datas, D, txyz, ttri, levels, varnorm, mask.empty() ? nullptr : &mask);
    const Triangles last_tri = make_mesh_from_icosa(levels.back().data_order).second;
    Mesh target(ctx, txyz, ttri);
    const std::string out = o.get("out");
    for (size_t s = 0; s < meshes.size(); ++s) {
        const std::string i = std::to_string(s);
        io::save_surface(out + "sphere-" + i + ".reg" + fmt.surf, res.sphere_regs[s], meshes[s].second);            // transform
        io::save_surface(out + "sphere-" + i + ".LR.reg" + fmt.surf, res.level_regs.back()[s], last_tri);           // saveSPH_reg
        Mesh moved(ctx, res.sphere_regs[s], meshes[s].second);
        save_data(out + "transformed_and_reprojected-" + i + fmt.data, txyz, metric_resample(moved, datas[s], target), D);  // save_transformed_data
    }
    if (o.has("verbose"))
        for (size_t k = 0; k < res.energies.size(); ++k) {
            std::cout << "level " << k + 1 << ": energies per iteration";
            for (double e : res.energies[k]) std::printf(" %.4f", e);
            std::cout << std::endl;
        }
    return 0;
}

}  // namespace

int main(int argc, char **argv) {
    try {
        const Options o = parse_command_line(argc, argv);
        if (o.has("help") || argc == 1) {
            usage();
            return 0;
        }
        if (o.has("verbose")) std::cout << "This is newMSM v1.0 on msm-mi355x." << std::endl;
        if (o.has("printoptions")) {
            std::cout << "The configuration grammar is the reference's (Mesh_registration::parse_reg_options, M/mesh_registration.cpp:459-784): one --key=value or --flag per "
                         "line, '#' comments; see include/msmhip_config.hpp for the keys." << std::endl;
            return 0;
        }
        if (o.get("out").empty()) throw Error(MSM_ERR_INVALID, "newmsm: --out is required");
        const Formats fmt = output_formats(o.has("format") ? o.get("format") : std::string("GIFTI"));
        const int device = o.has("device") ? std::atoi(o.get("device").c_str()) : 0;
        return o.has("groupwise") ? run_groupwise(o, fmt, device) : run_pairwise(o, fmt, device);
    } catch (const std::exception &e) {  // MeshregException: the message, exit status 1 (CLI/newmsm.cpp:62-68)
        std::cerr << e.what() << std::endl;
        return 1;
    }
}
