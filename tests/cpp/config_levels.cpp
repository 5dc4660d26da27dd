// config_levels.cpp -- include/msmhip_config.hpp as a compiled program: config_levels <config file> <D> prints the level schedule (or the error)
// as one JSON line, for comparison with newmsm_amd/config.py (tests/test_cpp_config.py).  Host logic only.
#include <cstdio>
#include <fstream>
#include <sstream>

#include "msmhip_config.hpp"

int main(int argc, char **argv) {
    if (argc != 3 && argc != 4) return 2;  // config_levels <file | NONE> <D> [anat]
    std::ifstream in(argv[1]);
    std::stringstream ss;
    ss << in.rdbuf();
    try {
        const msmhip::Config c = msmhip::parse_config(ss.str(), std::string(argv[1]) == "NONE");  // NONE: no --conf given
        bool vn = false;
        std::vector<std::pair<int, std::string>> skipped;
        const auto levels = msmhip::levels_from_config(c, std::atoi(argv[2]), &vn, &skipped, argc == 4);
        std::printf("{\"varnorm\": %s, \"skipped\": [", vn ? "true" : "false");
        for (size_t i = 0; i < skipped.size(); ++i) std::printf("%s[%d, \"%s\"]", i ? ", " : "", skipped[i].first, skipped[i].second.c_str());
        std::printf("], \"levels\": [");
        for (size_t i = 0; i < levels.size(); ++i) {
            const msmhip::LevelSpec &l = levels[i];
            const msmhip::LevelOptions &o = l.options;
            std::printf("%s{\"data_order\": %d, \"cp_order\": %d, \"sg_order\": %d, \"sigma_in\": %.17g, \"sigma_ref\": %.17g, \"iters\": %d, \"mciters\": %d, \"mcparam\": %.17g, "
                        "\"kind\": %d, \"simmeasure\": %d, \"rmode\": %d, \"rescale_labels\": %s, \"optimiser\": \"%s\", \"lambda_\": %.17g, \"mu\": %.17g, \"kappa\": %.17g, "
                        "\"k_exp\": %.17g, \"rexp\": %.17g, \"range_\": %.17g, \"percentile\": %.17g, \"anat_order\": %d}",
                        i ? ", " : "", l.data_order, l.cp_order, o.sg_order, l.sigma_in, l.sigma_ref, o.iters, o.mciters, o.mcparam, o.cost.kind, o.cost.simmeasure,
                        o.cost.regularisermode, o.rescale_labels ? "true" : "false", o.fusion ? "fusion" : (o.pairwise ? "fastpd" : "mcmc"), o.cost.lambda,
                        o.cost.shearmodulus, o.cost.bulkmodulus, o.cost.kexponent, o.cost.exponent, o.cost.range, o.cost.percentile, o.anat_order);
        }
        std::printf("]}\n");
        return 0;
    } catch (const msmhip::ConfigError &e) {
        std::string m = e.what();
        for (char &ch : m)
            if (ch == '"') ch = '\'';
        std::printf("{\"error\": \"%s\"}\n", m.c_str());
        return 0;
    }
}
