// registration_bench.cpp -- a whole pairwise registration driven by the C++ host code (include/msmhip_registration.hpp: run_multiresolutions over
// include/msmhip.hpp over the C ABI), schedule read from a configuration file through the reference's grammar (include/msmhip_config.hpp): no
// Python between the optimiser's loop and the library.  north_star: "host code stays C++ and calls HIP through a thin C-ABI"; bench.py runs this
// program as a child process and reports it as registration_msmall_cpp / registration_fusion_cpp next to the Python-driven objects.
//
//   registration_bench <in.bag> <out.bag> <config file> [runs]
//
// in.bag  (newmsm_amd/bag.py):  orders i4 [sphere order, D], in_data f8 D x V, ref_data f8 D x V, [iters i4: iterations per level],
//          [in_anat, ref_anat f8 V x 3: the anatomical surfaces of a --regoption=5 (aMSM) configuration]
// out.bag: sphere_reg f8 3 x V (AoS), labelings i4 (all iterations, level after level), nodes i4 (control points per labeling), energies f8,
//          move_kernel_ms f8 (one per fusion move of the extra timed run)
// stdout: one JSON line -- wall_s, path_s (everything but the stand-in solve), phases_s, calls per phase -- of the LAST of `runs` runs (default 2:
//         the first pays for allocations), and the per-move figures.
// Matches: Mesh_registration::run_multiresolutions M/mesh_registration.cpp:30-50, run_discrete_opt :164-232, Fusion::optimize I/Fusion/Fusion.h:136-229.
#include <chrono>
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <string>

#include "msmhip_config.hpp"

using namespace msmhip;

static std::map<std::string, std::vector<double>> F;
static std::map<std::string, std::vector<int32_t>> I;

static void read_bag(const char *path) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error(std::string("cannot open ") + path);
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty()) continue;
        std::istringstream hs(line);
        std::string name, dtype;
        size_t n;
        hs >> name >> dtype >> n;
        if (dtype == "f8") {
            F[name].resize(n);
            in.read(reinterpret_cast<char *>(F[name].data()), (std::streamsize)(n * 8));
        } else {
            I[name].resize(n);
            in.read(reinterpret_cast<char *>(I[name].data()), (std::streamsize)(n * 4));
        }
    }
}
template <class T>
static void put(std::ofstream &out, const std::string &name, const char *dtype, const std::vector<T> &v) {
    out << name << " " << dtype << " " << v.size() << "\n";
    out.write(reinterpret_cast<const char *>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: registration_bench <in.bag> <out.bag> <config file> [runs]\n");
        return 2;
    }
    try {
        read_bag(argv[1]);
        const int order = I.at("orders").at(0), D = I.at("orders").at(1), runs = argc > 4 ? std::atoi(argv[4]) : 2;
        std::ifstream cf(argv[3]);
        if (!cf) throw std::runtime_error(std::string("cannot open ") + argv[3]);
        std::stringstream ss;
        ss << cf.rdbuf();
        bool varnorm = false;
        std::vector<std::pair<int, std::string>> skipped;
        const bool anat = F.count("in_anat") && F.count("ref_anat");
        const Points *in_anat = anat ? &F["in_anat"] : nullptr, *ref_anat = anat ? &F["ref_anat"] : nullptr;
        std::vector<LevelSpec> levels = levels_from_config(parse_config(ss.str()), D, &varnorm, &skipped, anat);
        if (I.count("iters"))
            for (size_t k = 0; k < levels.size() && k < I["iters"].size(); ++k) levels[k].options.iters = I["iters"][k];
        auto [xyz, tri] = make_mesh_from_icosa(order);
        Context ctx(0);
        MultiresResult res;
        PhaseClock clock;
        double wall = 0.0;
        for (int r = 0; r < runs; ++r) {
            clock = PhaseClock();
            const auto t0 = std::chrono::steady_clock::now();
            res = run_multiresolutions(ctx, xyz, tri, F.at("in_data"), xyz, tri, F.at("ref_data"), D, levels, varnorm, &clock, in_anat, ref_anat);
            wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        // once more with HIP events around every fusion move's kernel (not the run that is reported: an event query per move)
        std::vector<double> kernel_ms;
        if (levels[0].options.fusion) {
            std::vector<LevelSpec> timed = levels;
            for (LevelSpec &lv : timed) lv.options.move_kernel_ms = &kernel_ms;
            const MultiresResult again = run_multiresolutions(ctx, xyz, tri, F.at("in_data"), xyz, tri, F.at("ref_data"), D, timed, varnorm, nullptr, in_anat, ref_anat);
            if (again.labelings != res.labelings) throw std::runtime_error("the timed run took different decisions");
        }
        std::ofstream out(argv[2], std::ios::binary);
        put(out, "sphere_reg", "f8", res.sphere_reg);
        std::vector<int32_t> lab, nodes;
        std::vector<double> energies;
        for (const auto &l : res.labelings) {
            lab.insert(lab.end(), l.begin(), l.end());
            nodes.push_back((int32_t)l.size());
        }
        for (const auto &e : res.energies) energies.insert(energies.end(), e.begin(), e.end());
        put(out, "labelings", "i4", lab);
        put(out, "nodes", "i4", nodes);
        put(out, "energies", "f8", energies);
        put(out, "move_kernel_ms", "f8", kernel_ms);
        double path = 0.0, ksum = 0.0;
        for (const auto &kv : clock.seconds)
            if (kv.first != "optimiser") path += kv.second;
        for (double k : kernel_ms) ksum += k;
        std::printf("{\"wall_s\": %.6f, \"path_s\": %.6f, \"runs\": %d, \"levels\": %zu, \"skipped_levels\": %zu, \"phases_s\": {", wall, path, runs, levels.size(), skipped.size());
        bool first = true;
        for (const auto &kv : clock.seconds) {
            std::printf("%s\"%s\": %.6f", first ? "" : ", ", kv.first.c_str(), kv.second);
            first = false;
        }
        std::printf("}, \"calls\": {");
        first = true;
        for (const auto &kv : clock.calls) {
            std::printf("%s\"%s\": %ld", first ? "" : ", ", kv.first.c_str(), kv.second);
            first = false;
        }
        const long moves = clock.calls.count("fusion_moves") ? clock.calls["fusion_moves"] : 0;
        std::printf("}, \"moves\": %ld, \"move_us_per_call\": %.3f, \"move_kernel_us\": %.3f, \"moves_timed\": %zu}\n", moves,
                    moves ? clock.seconds["fusion_moves"] / moves * 1e6 : 0.0, kernel_ms.empty() ? 0.0 : ksum / kernel_ms.size() * 1e3, kernel_ms.size());
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "registration_bench failed: %s\n", e.what());
        return 1;
    }
}
