// io_roundtrip.cpp -- include/msmhip_io.hpp as a compiled program (g++ -lz -lexpat, no GPU, no Python in the loop):
//   io_roundtrip surf   <in> <out>    load_surface(in)  -> save_surface(out)
//   io_roundtrip metric <in> <out>    load_metric(in)   -> save_metric(out)
//   io_roundtrip dump   <in> <out>    every array of a GIFTI file as text: intent, dims, values (%.17g)
// tests/test_cpp_io.py compares the outputs with newmsm_amd/meshio.py byte for byte.
#include <cstdio>
#include <string>

#include "msmhip_io.hpp"

int main(int argc, char **argv) {
    if (argc != 4) return 2;
    const std::string mode = argv[1], in = argv[2], out = argv[3];
    try {
        if (mode == "surf") {
            auto [xyz, tri] = msmhip::io::load_surface(in);
            msmhip::io::save_surface(out, xyz, tri);
        } else if (mode == "metric") {
            int D = 0;
            const std::vector<double> data = msmhip::io::load_metric(in, &D);
            msmhip::io::save_metric(out, data, D);
        } else if (mode == "dump") {
            FILE *f = std::fopen(out.c_str(), "w");
            if (!f) return 3;
            for (const auto &a : msmhip::io::read_gifti(in)) {
                std::fprintf(f, "%s", a.intent.c_str());
                for (const auto d : a.dims) std::fprintf(f, " %lld", (long long)d);
                std::fprintf(f, "\n");
                for (const double v : a.values) std::fprintf(f, "%.17g ", v);
                std::fprintf(f, "\n");
            }
            std::fclose(f);
        } else {
            return 2;
        }
        std::puts("ok");
        return 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "io_roundtrip: %s\n", e.what());
        return 1;
    }
}
