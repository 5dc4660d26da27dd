// level_driver.cpp -- run_discrete_opt of include/msmhip_registration.hpp as a compiled program (g++ + libmsmhip.so, no
// Python in the loop), for comparison with newmsm_amd/registration.py on the same inputs (tests/test_cpp_host.py).
//
//   level_driver <in.bin> <out.bin>      file format: host_mirror.cpp
#include <cstdio>
#include <fstream>
#include <map>
#include <sstream>
#include <string>

#include "msmhip_registration.hpp"

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
    if (argc != 3) return 2;
    try {
        read_bag(argv[1]);
        const int data_order = I["orders"][0], cp_order = I["orders"][1], D = I["orders"][2];
        auto [xyz, tri] = make_mesh_from_icosa(data_order);
        Context ctx(0);
        LevelOptions o;
        o.iters = I["orders"][3];
        o.mciters = I["orders"][4];
        o.seed = (uint64_t)I["orders"][5];
        o.rescale_labels = I["orders"][6] != 0;
        o.mcparam = F["params"][0];
        o.cost.lambda = F["params"][1];
        o.cost.kind = D > 1 ? MSM_COST_MULTIVARIATE : MSM_COST_UNIVARIATE;
        if (I["orders"].size() > 7 && I["orders"][7] != 0) {  // the fusion-driven loop over the triclique classes with the HCP regulariser
            o.fusion = true;
            o.cost.kind = D > 1 ? MSM_COST_HO_MULTIVARIATE : MSM_COST_HO_UNIVARIATE;
            o.cost.shearmodulus = 0.4, o.cost.bulkmodulus = 1.6, o.cost.kexponent = 2.0, o.cost.exponent = 2.0;
        }
        if (I["orders"].size() > 8 && I["orders"][8] != 0) {  // --regoption=1 as --dopt=FastPD drives it: unary + pair tables, stand-in solve
            o.pairwise = true;
            o.cost.regularisermode = 1;
        }
        const LevelResult r = run_discrete_opt(ctx, xyz, tri, F["ref_feat"], xyz, tri, F["src_feat"], D, xyz, cp_order, o);
        std::ofstream out(argv[2], std::ios::binary);
        put(out, "sph_reg", "f8", r.sph_reg);
        put(out, "cpgrid", "f8", r.cpgrid);
        put(out, "energies", "f8", r.energies);
        std::vector<int32_t> lab;
        for (const auto &l : r.labelings) lab.insert(lab.end(), l.begin(), l.end());
        put(out, "labelings", "i4", lab);
        std::puts("ok");
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "level_driver failed: %s\n", e.what());
        return 1;
    }
}
