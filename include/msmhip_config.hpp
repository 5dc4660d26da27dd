// msmhip_config.hpp -- the reference's configuration-file grammar (`newmsm --conf=<file>`) -> the level schedule of msmhip_registration.hpp.
//
//     Mesh_registration::parse_reg_options         M/mesh_registration.cpp:459-784   keys, types, defaults, consistency checks and messages
//     Mesh_registration::fix_parameters_for_level  M/mesh_registration.cpp:786-817   what a level hands to the model / cost function
//     NonLinearSRegDiscreteModel::set_parameters / initialize_cost_function   M/DiscreteModel.cpp:26-60   the cost class of a level
//
// Header only, C++17, host logic (no GPU call).  One `--key=value` or `--flag` per line, `#` starts a comment, blank lines are skipped.  Values
// whose option type is `float` in the reference (--lambda, --sigma_in, --sigma_ref, --cutthr, --shearmod, --bulkmod, --k_exponent, --regexp,
// --cprange, --stepsize, --gradsampling, --mcparam, --percentile) pass through a float on their way to double, as Utilities::Option<float> /
// std::vector<float> make them there: --lambda=0.0075 reaches the cost function as 0.007499999832361937.  newmsm_amd/config.py is the same
// parser in Python; tests/test_cpp_config.py checks that the two agree.
//
// Reported instead of silently dropped: AFFINE / RIGID levels (the affine stage is outside the path: listed in `skipped`), --IN / --INc (FSL's
// histogram matching is not in the reference tree), --excl.  --regoption=5 (aMSM) needs the anatomical surfaces (command line: --inanat / --refanat):
// levels_from_config(..., anat = true) says the caller has them.
#ifndef MSMHIP_CONFIG_HPP
#define MSMHIP_CONFIG_HPP

#include <cstdlib>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "msmhip_registration.hpp"

namespace msmhip {

struct Config {  // the members parse_reg_options fills (M/mesh_registration.h:100-145)
    std::vector<std::string> opt;
    std::vector<int> simval, it, datagrid, CPgrid, SGgrid, anatgrid, mciters;
    std::vector<double> sigma_in, sigma_ref, lambda, cutthr{0.0, (double)0.0001f};
    int regoption = 1, numthreads = 1;
    std::string dopt = "FastPD";
    double shearmod = (double)0.4f, bulkmod = (double)1.6f, k_exponent = 2.0, regexp = 2.0, cprange = 1.0, stepsize = (double)0.01f, gradsampling = 0.5,
           mcparam = (double)0.8f, percentile = 0.75;
    bool triclique = false, patchwise = false, fixnan = false, rescaleL = false, IN = false, INc = false, VN = false, excl = false;
};

struct ConfigError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// (LevelSpec -- data grid, control grid, smoothing and the LevelOptions of one level -- lives in msmhip_registration.hpp, next to the loop that runs it)

namespace detail {
inline std::string trim(const std::string &s) {
    const size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
inline std::vector<std::string> split_commas(const std::string &s) {
    std::vector<std::string> out;
    std::string item;
    std::istringstream in(s);
    while (std::getline(in, item, ',')) out.push_back(trim(item));
    if (!s.empty() && s.back() == ',') out.push_back("");
    return out;
}
inline double to_double(const std::string &key, const std::string &v) {
    char *end = nullptr;
    const double d = std::strtod(v.c_str(), &end);
    if (v.empty() || *end) throw ConfigError("cannot read the value of --" + key + ": '" + v + "'");
    return d;
}
inline double to_float(const std::string &key, const std::string &v) { return (double)(float)to_double(key, v); }
inline int to_int(const std::string &key, const std::string &v) {
    char *end = nullptr;
    const long d = std::strtol(v.c_str(), &end, 10);
    if (v.empty() || *end) throw ConfigError("cannot read the value of --" + key + ": '" + v + "'");
    return (int)d;
}
}  // namespace detail

// text: the contents of the configuration file; no_config: no --conf was given (the reference branches on the file NAME being empty,
// M/mesh_registration.cpp:627 -- an empty file is not "no config": it yields zero levels)
inline Config parse_config(const std::string &text, bool no_config = false) {
    Config c;
    std::set<std::string> seen;
    std::istringstream in(text);
    std::string line;
    int lineno = 0;
    while (std::getline(in, line)) {
        ++lineno;
        line = detail::trim(line.substr(0, line.find('#')));
        if (line.empty()) continue;
        const std::string where = "line " + std::to_string(lineno) + ": ";
        if (line.compare(0, 2, "--") != 0) throw ConfigError(where + "expected --key=value or --flag, got '" + line + "'");
        const size_t eq = line.find('=');
        const std::string key = detail::trim(line.substr(2, eq == std::string::npos ? std::string::npos : eq - 2));
        const std::string value = eq == std::string::npos ? std::string() : detail::trim(line.substr(eq + 1));
        const std::map<std::string, bool *> flags = {{"triclique", &c.triclique}, {"patchwise", &c.patchwise}, {"fixnan", &c.fixnan}, {"rescaleL", &c.rescaleL},
                                                    {"IN", &c.IN},               {"INc", &c.INc},             {"VN", &c.VN},         {"excl", &c.excl}};
        const std::map<std::string, std::vector<int> *> int_lists = {{"simval", &c.simval},     {"it", &c.it},           {"datagrid", &c.datagrid}, {"CPgrid", &c.CPgrid},
                                                                     {"SGgrid", &c.SGgrid},     {"anatgrid", &c.anatgrid}, {"mciters", &c.mciters}};
        const std::map<std::string, std::vector<double> *> float_lists = {{"sigma_in", &c.sigma_in}, {"sigma_ref", &c.sigma_ref}, {"lambda", &c.lambda}, {"cutthr", &c.cutthr}};
        const std::map<std::string, double *> floats = {{"shearmod", &c.shearmod}, {"bulkmod", &c.bulkmod},   {"k_exponent", &c.k_exponent},     {"regexp", &c.regexp},
                                                        {"cprange", &c.cprange},   {"stepsize", &c.stepsize}, {"gradsampling", &c.gradsampling}, {"mcparam", &c.mcparam},
                                                        {"percentile", &c.percentile}};
        if (flags.count(key)) {
            if (eq != std::string::npos) throw ConfigError(where + "--" + key + " takes no argument");
            *flags.at(key) = true;
        } else if (int_lists.count(key) || float_lists.count(key) || floats.count(key) || key == "opt" || key == "regoption" || key == "numthreads" || key == "dopt") {
            if (eq == std::string::npos || value.empty()) throw ConfigError(where + "--" + key + " requires an argument");
            try {
                if (int_lists.count(key)) {
                    int_lists.at(key)->clear();
                    for (const std::string &v : detail::split_commas(value)) int_lists.at(key)->push_back(detail::to_int(key, v));
                } else if (float_lists.count(key)) {
                    float_lists.at(key)->clear();
                    for (const std::string &v : detail::split_commas(value)) float_lists.at(key)->push_back(detail::to_float(key, v));
                } else if (floats.count(key)) {
                    *floats.at(key) = detail::to_float(key, value);
                } else if (key == "opt") {
                    c.opt = detail::split_commas(value);
                } else if (key == "regoption") {
                    c.regoption = detail::to_int(key, value);
                } else if (key == "numthreads") {
                    c.numthreads = detail::to_int(key, value);
                } else {
                    c.dopt = value;
                }
            } catch (const ConfigError &e) {
                throw ConfigError(where + e.what());
            }
        } else {
            throw ConfigError(where + "unrecognised option --" + key);
        }
        seen.insert(key);
    }
    auto set = [&](const char *k) { return seen.count(k) != 0; };
    if (no_config) {  // no config: the sulc configuration of September 2014 (M/mesh_registration.cpp:629-642)
        c.opt = {"RIGID", "DISCRETE", "DISCRETE", "DISCRETE"};
        c.lambda = {0.0, (double)0.1f, (double)0.2f, (double)0.3f};
        c.simval = {1, 2, 2, 2};
        c.sigma_in = {2.0, 2.0, 3.0, 2.0};
        c.sigma_ref = {2.0, 2.0, 1.5, 1.0};
        c.it = {50, 3, 3, 3};
        c.CPgrid = {0, 2, 3, 4};
        c.anatgrid = {0, 4, 5, 6};
        c.datagrid = {4, 4, 5, 6};
        c.SGgrid = {0, 4, 5, 6};
    } else {
        const size_t n = c.opt.size();
        if (!set("simval")) c.simval.assign(n, 2);
        for (int &v : c.simval)
            if (v == 3) v = 2;  // NMI was removed: Pearson's correlation instead (:648-653)
        if (!set("it")) c.it.assign(n, 3);
        if (!set("sigma_in")) c.sigma_in.assign(n, 2.0);
        if (!set("sigma_ref")) c.sigma_ref = c.sigma_in;
        if (!set("datagrid")) c.datagrid.assign(n, 5);
        if (!set("CPgrid")) {
            c.CPgrid.resize(n);
            for (size_t i = 0; i < n; ++i) c.CPgrid[i] = 2 + (int)i;
        }
        if (!set("anatgrid")) {
            c.anatgrid.assign(n, 2);
            for (size_t i = 0; i < n && i < c.CPgrid.size(); ++i) c.anatgrid[i] = c.CPgrid[i] + 2;
        }
        if (!set("SGgrid")) {
            c.SGgrid.assign(n, 0);
            for (size_t i = 0; i < n && i < c.CPgrid.size(); ++i) c.SGgrid[i] = c.CPgrid[i] + 2;
        }
    }
    const size_t n = c.opt.size();
    if (!set("mciters")) c.mciters.assign(n, 100000);
    if (c.dopt == "FastPD") c.regoption = 1;  // :684
    if (c.regoption > 1 && c.dopt == "FastPD") throw ConfigError("MeshREG ERROR:: you cannot run higher order clique regularisers with fastPD ");
    if (c.cutthr.size() != 2) throw ConfigError("MeshREG ERROR:: the cut threshold does not contain a limit for upper and lower threshold (too few inputs)");
    const std::pair<size_t, const char *> lists[] = {{c.simval.size(), "--simval"},       {c.it.size(), "--it"},         {c.sigma_in.size(), "--sigma_in"},
                                                     {c.sigma_ref.size(), "--sigma_ref"}, {c.lambda.size(), "--lambda"}, {c.datagrid.size(), "--datagrid"},
                                                     {c.CPgrid.size(), "--CPgrid"},       {c.SGgrid.size(), "--SGres"}};
    for (const auto &l : lists)
        if (l.first != n) throw ConfigError(std::string("MeshREG ERROR:: config file parameter list lengths are inconsistent: ") + l.second);
    if (c.patchwise && c.triclique) throw ConfigError("Cannot use patchwise and triclique options together. Choose one.");
    if (c.percentile < 0.0 + 1e-8 || c.percentile > 1.0 - 1e-8) throw ConfigError("Percentile must be between 0 and 1.");
    return c;
}

// the DISCRETE levels of `c` for data with D feature rows; skipped (optional): index and method of the levels that are not DISCRETE
// anat: the caller has the anatomical surfaces a --regoption=5 (aMSM) run needs (they come from the command line: --inanat / --refanat)
inline std::vector<LevelSpec> levels_from_config(const Config &c, int D, bool *varnorm = nullptr, std::vector<std::pair<int, std::string>> *skipped = nullptr,
                                                 bool anat = false) {
    if (c.IN || c.INc) throw ConfigError("--IN / --INc (histogram matching through FSL's MISCMATHS::Histogram, M/reg_tools.cpp:745-802) is not available");
    if (c.excl) throw ConfigError("--excl (exclusion masks from the cut thresholds) is not wired into the level loop");
    if (c.regoption == 4)  // M/mesh_registration.cpp:101-102
        throw ConfigError("--regoption 4 has been removed from newMSM. Use --regoption 3 for spherical mesh regularisation or --regoption 5 for anatomical mesh "
                          "regularisation.");
    if (c.regoption == 5 && !anat)  // :103-104
        throw ConfigError("--regoption 5 requires anatomical meshes. Use --regoption 3 for spherical mesh regularisation or provide anatomical meshes.");
    int kind;
    if (D > 1) kind = c.patchwise ? MSM_COST_PATCHWISE : (c.triclique ? MSM_COST_HO_MULTIVARIATE : MSM_COST_MULTIVARIATE);  // M/DiscreteModel.cpp:44-58
    else kind = c.triclique ? MSM_COST_HO_UNIVARIATE : MSM_COST_UNIVARIATE;
    if (c.dopt != "HOCR" && c.dopt != "MCMC" && c.dopt != "FastPD") throw ConfigError("Unrecognized optimiser");  // M/mesh_registration.cpp:202
    if (c.dopt != "FastPD" && c.regoption == 1)
        throw ConfigError("--regoption=1 (pairwise regulariser) is driven by FastPD only in the reference; Fusion / MCMC read triplets");
    if (varnorm) *varnorm = c.VN;
    std::vector<LevelSpec> levels;
    for (size_t i = 0; i < c.opt.size(); ++i) {
        if (c.opt[i] != "DISCRETE") {
            if (skipped) skipped->emplace_back((int)i, c.opt[i]);
            continue;
        }
        LevelSpec lv;
        lv.data_order = c.datagrid[i], lv.cp_order = c.CPgrid[i], lv.sigma_in = c.sigma_in[i], lv.sigma_ref = c.sigma_ref[i];
        LevelOptions &o = lv.options;
        o.sg_order = c.SGgrid[i], o.iters = c.it[i], o.mciters = c.mciters[i], o.mcparam = c.mcparam, o.rescale_labels = c.rescaleL;
        o.fusion = c.dopt == "HOCR", o.pairwise = c.dopt == "FastPD";
        o.anat_order = i < c.anatgrid.size() ? c.anatgrid[i] : c.CPgrid[i] + 2;
        o.cost.kind = kind, o.cost.simmeasure = c.simval[i], o.cost.regularisermode = c.regoption, o.cost.lambda = c.lambda[i];
        o.cost.shearmodulus = c.shearmod, o.cost.bulkmodulus = c.bulkmod, o.cost.kexponent = c.k_exponent, o.cost.exponent = c.regexp;
        o.cost.range = c.cprange, o.cost.percentile = c.percentile;
        levels.push_back(lv);
    }
    return levels;
}

}  // namespace msmhip

#endif  // MSMHIP_CONFIG_HPP
