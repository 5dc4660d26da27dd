// group_driver.cpp -- run_group_multiresolutions of include/msmhip_group_registration.hpp as a compiled program (g++ + libmsmhip.so, no Python in
// the loop), for comparison with newmsm_amd/group_registration.py on the same inputs (tests/test_cpp_host.py).
//
//   group_driver <in.bin> <out.bin> [config]     file format: host_mirror.cpp; config: a newmsm configuration file -- the levels then come from it
//                                                (group_levels_from_config) instead of from level_orders / level_params
// in:  sizes [S, D, levels, varnorm, fixnan, masked], per subject mesh<i>_xyz / mesh<i>_tri / data<i>, template_xyz / template_tri, mask,
//      level_orders [data_order, cp_order, sg_order, iters, simval] per level, level_params [sigma_in, lambda] per level
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <string>

#include "msmhip_group_registration.hpp"

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
    if (argc != 3 && argc != 4) return 2;
    try {
        read_bag(argv[1]);
        const int S = I["sizes"][0], D = I["sizes"][1], nlevels = I["sizes"][2];
        const bool varnorm = I["sizes"][3] != 0, fixnan = I["sizes"][4] != 0, masked = I["sizes"][5] != 0;
        std::vector<std::pair<Points, Triangles>> meshes;
        std::vector<Matrix> datas;
        for (int s = 0; s < S; ++s) {
            const std::string k = std::to_string(s);
            meshes.emplace_back(F["mesh" + k + "_xyz"], I["mesh" + k + "_tri"]);
            datas.push_back(F["data" + k]);
        }
        std::vector<GroupLevelSpec> levels((size_t)nlevels);
        for (int l = 0; l < nlevels; ++l) {
            GroupLevelSpec &lv = levels[(size_t)l];
            lv.data_order = I["level_orders"][5 * l], lv.cp_order = I["level_orders"][5 * l + 1];
            lv.options.sg_order = I["level_orders"][5 * l + 2], lv.options.iters = I["level_orders"][5 * l + 3];
            lv.options.cost.simmeasure = I["level_orders"][5 * l + 4];
            lv.sigma_in = F["level_params"][2 * l];
            lv.options.cost.lambda = F["level_params"][2 * l + 1];
            lv.options.cost.fixnan = fixnan;
        }
        bool vn = varnorm;
        if (argc == 4) {
            std::ifstream cf(argv[3]);
            if (!cf) throw std::runtime_error(std::string("cannot open ") + argv[3]);
            std::stringstream text;
            text << cf.rdbuf();
            levels = group_levels_from_config(parse_config(text.str()), &vn);
        }
        Context ctx(0);
        PhaseClock clock;
        const GroupMultiresResult r = run_group_multiresolutions(ctx, meshes, datas, D, F["template_xyz"], I["template_tri"], levels, vn, masked ? &F["mask"] : nullptr, &clock);
        std::ofstream out(argv[2], std::ios::binary);
        for (int s = 0; s < S; ++s) {
            put(out, "sphere_reg" + std::to_string(s), "f8", r.sphere_regs[(size_t)s]);
            put(out, "level_reg" + std::to_string(s), "f8", r.level_regs.back()[(size_t)s]);
        }
        std::vector<double> energies;
        for (const auto &e : r.energies) energies.insert(energies.end(), e.begin(), e.end());
        put(out, "energies", "f8", energies);
        std::vector<int32_t> lab;
        for (const auto &l : r.labelings) lab.insert(lab.end(), l.begin(), l.end());
        put(out, "labelings", "i4", lab);
        std::puts("ok");
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "group_driver failed: %s\n", e.what());
        return 1;
    }
}
