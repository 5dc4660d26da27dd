// rccl_group.cpp -- include/msmhip_rccl.hpp driven from a C++ host: the sharded gMSM set-up, a gathered label step and the template
// update through RCCL, against the unsharded entry points of the C ABI (exact equality).  One process = one rank; with a single
// GPU in the box the communicator has one rank, which still takes every call through RCCL (bootstrap, all-gather, grouped
// send / receive, all-reduce).  The exchange across ranks is the same code as newmsm_amd/dist.py's, which the two-rank tests cover.
// build: hipcc -std=c++17 -I include tests/cpp/rccl_group.cpp -L newmsm_amd -lmsmhip -lrccl
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "msmhip_rccl.hpp"

using msmhip::rccl::check_hip;
using msmhip::rccl::check_msm;

struct Problem {
    msm_ctx *ctx = nullptr;
    msm_group *g = nullptr;
    std::vector<msm_mesh *> meshes;
    int S = 3, L = 0, D = 1;
    int32_t V = 0, T = 0, N = 0, Tc = 0;
    std::vector<double> spheres;  // S x V x 3, the "registered" spheres of the template update
    std::vector<double> feats;    // S x D x V
};

static Problem make_problem(msm_ctx *ctx) {
    Problem p;
    p.ctx = ctx;
    check_msm(msm_icosphere_counts(3, &p.V, &p.T), "counts");
    check_msm(msm_icosphere_counts(1, &p.N, &p.Tc), "counts");
    std::vector<double> xyz(3 * (size_t)p.V), cxyz(3 * (size_t)p.N);
    std::vector<int32_t> tri(3 * (size_t)p.T), ctri(3 * (size_t)p.Tc);
    check_msm(msm_icosphere(3, 100.0, xyz.data(), tri.data()), "icosphere");
    check_msm(msm_icosphere(1, 100.0, cxyz.data(), ctri.data()), "icosphere");
    std::vector<double> maxsep(p.N);
    double mvd = 0;
    check_msm(msm_cp_spacings(cxyz.data(), ctri.data(), p.N, p.Tc, maxsep.data(), &mvd), "spacings");
    std::vector<double> samples(3 * 256), bary(3 * 256);
    int32_t ns = 0, nb = 0;
    check_msm(msm_label_sampling_grid(3, 0.5 * mvd, 0, 256, samples.data(), &ns, bary.data(), &nb), "sampling grid");
    std::vector<double> labels(3 * (size_t)ns);
    for (int a = 0; a < 3; ++a)
        for (int i = 0; i < ns; ++i) labels[(size_t)a * ns + i] = samples[(size_t)a * 256 + i];
    p.L = ns;
    msm_group_params prm{};
    prm.simmeasure = 2, prm.lambda = 0.2, prm.mu = 0.1, prm.kappa = 10.0, prm.k_exp = 2.0, prm.rexp = 2.0, prm.range = 1.0;
    p.g = msm_group_create(ctx, &prm, p.S);
    if (!p.g) throw msmhip::rccl::Error(msm_last_error());
    msm_mesh *tm = msm_mesh_create(ctx, xyz.data(), p.V, tri.data(), p.T);
    p.meshes.push_back(tm);
    check_msm(msm_group_set_template(p.g, tm, nullptr), "template");
    check_msm(msm_group_set_controlgrid(p.g, cxyz.data(), ctri.data(), p.N, p.Tc), "control grid");
    auto rotated = [](const std::vector<double> &in, int n, double deg) {
        std::vector<double> out(in.size());
        const double c = std::cos(deg * M_PI / 180.0), s = std::sin(deg * M_PI / 180.0);
        for (int i = 0; i < n; ++i) {
            const double x = in[i], y = in[(size_t)n + i], z = in[2 * (size_t)n + i];
            out[i] = c * x - s * y, out[(size_t)n + i] = s * x + c * y, out[2 * (size_t)n + i] = z;
        }
        return out;
    };
    p.spheres.resize((size_t)p.S * p.V * 3);
    p.feats.resize((size_t)p.S * p.D * p.V);
    for (int s = 0; s < p.S; ++s) {
        std::vector<double> f(p.V);
        for (int v = 0; v < p.V; ++v) f[v] = std::sin(0.05 * xyz[v] + 0.3 * s) + std::cos(0.03 * xyz[(size_t)p.V + v]) + 0.01 * xyz[2 * (size_t)p.V + v];
        std::copy(f.begin(), f.end(), p.feats.begin() + (size_t)s * p.V);
        msm_mesh *m = msm_mesh_create(ctx, xyz.data(), p.V, tri.data(), p.T);
        p.meshes.push_back(m);
        check_msm(msm_group_set_subject(p.g, s, m, f.data(), 1), "subject (original)");
        const std::vector<double> moved = rotated(xyz, p.V, 1.0 + 0.7 * s);
        check_msm(msm_mesh_update_coords(m, moved.data()), "update coords");
        check_msm(msm_group_set_subject(p.g, s, m, f.data(), 1), "subject (moved)");
        const std::vector<double> cm = rotated(cxyz, p.N, 1.0 + 0.7 * s);
        check_msm(msm_group_reset_cpgrid(p.g, s, cm.data()), "reset control grid");
        for (int v = 0; v < p.V; ++v)
            for (int a = 0; a < 3; ++a) p.spheres[((size_t)s * p.V + v) * 3 + a] = moved[(size_t)a * p.V + v];
    }
    check_msm(msm_group_set_labels(p.g, labels.data(), p.L), "labels");
    return p;
}

static void destroy(Problem &p) {
    msm_group_destroy(p.g);
    for (msm_mesh *m : p.meshes) msm_mesh_destroy(m);
}

int main() {
    try {
        if (msm_device_count() < 1) {
            std::fprintf(stderr, "no GPU\n");
            return 2;
        }
        msm_ctx *ctx = msm_ctx_create(0);
        if (!ctx) throw msmhip::rccl::Error(msm_last_error());
        Problem a = make_problem(ctx), b = make_problem(ctx);
        // the unsharded reference: plain ABI calls (the pair list in the layout sharded_group_setup gives every launched run, whatever its rank count)
        check_msm(msm_group_set_pair_layout(a.g, 1), "msm_group_set_pair_layout");
        check_msm(msm_group_setup(a.g), "msm_group_setup");
        int32_t nodes = 0, P = 0, T = 0;
        check_msm(msm_group_sizes(a.g, &nodes, &P, &T), "sizes");
        std::vector<int32_t> labeling(nodes);
        for (int i = 0; i < nodes; ++i) labeling[i] = (7 * i + 3) % a.L;
        std::vector<double> quads(4 * (size_t)P), octets(8 * (size_t)T);
        check_msm(msm_group_fusion_move(a.g, labeling.data(), 2, quads.data(), octets.data()), "msm_group_fusion_move");
        // the same through RCCL
        ncclUniqueId id;
        msmhip::rccl::check_nccl(ncclGetUniqueId(&id), "ncclGetUniqueId");
        msmhip::rccl::Comm comm(0, 1, id, (hipStream_t)msm_ctx_stream(ctx));
        const std::vector<int32_t> mine = msmhip::rccl::sharded_group_setup(b.g, comm);
        if ((int)mine.size() != b.S) throw msmhip::rccl::Error("a one-rank shard must hold every subject");
        const size_t n = 4 * (size_t)P + 8 * (size_t)T;
        double *send = nullptr, *recv = nullptr;
        check_hip(hipMalloc((void **)&send, n * sizeof(double)), "hipMalloc");
        check_hip(hipMalloc((void **)&recv, n * sizeof(double)), "hipMalloc");
        msmhip::rccl::gather_label_step(b.g, labeling.data(), 2, send, recv, 0, comm);
        std::vector<double> got(n);
        check_hip(hipMemcpy(got.data(), recv, n * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy");
        size_t bad = 0;
        for (size_t i = 0; i < 4 * (size_t)P; ++i) bad += !(got[i] == quads[i] || (got[i] != got[i] && quads[i] != quads[i]));
        for (size_t i = 0; i < 8 * (size_t)T; ++i) bad += !(got[4 * (size_t)P + i] == octets[i]);
        // template update: one all-reduce; compared with the direct mean
        const msmhip::rccl::TemplateUpdate up = msmhip::rccl::group_template_update(b.spheres.data(), b.S, b.V, b.feats.data(), b.D, comm);
        double worst = 0.0;
        for (int v = 0; v < b.V; ++v) {
            double m[3] = {0, 0, 0};
            for (int s = 0; s < b.S; ++s)
                for (int k = 0; k < 3; ++k) m[k] += b.spheres[((size_t)s * b.V + v) * 3 + k] / b.S;
            const double len = std::sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
            for (int k = 0; k < 3; ++k) worst = std::max(worst, std::fabs(up.sphere[3 * (size_t)v + k] - m[k] * 100.0 / len));
        }
        (void)hipFree(send);
        (void)hipFree(recv);
        std::printf("{\"pairs\": %d, \"triplets\": %d, \"mismatches\": %zu, \"template_max_err\": %.3e, \"n_subjects\": %lld}\n", P, T, bad, worst, up.n_subjects);
        destroy(a);
        destroy(b);
        msm_ctx_destroy(ctx);
        return (bad == 0 && worst < 1e-9 && up.n_subjects == 3) ? 0 : 1;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "rccl_group: %s\n", e.what());
        return 3;
    }
}
