// cost_cliques.cpp -- triplet / pairwise clique costs of the discrete cost function (placeholder until
// the clique kernels land; every symbol of msmhip.h must exist).
#include "internal.hpp"

using namespace msm;

extern "C" {

int msm_cost_triplet_batch(msm_cost *, const int32_t *, const int32_t *, const int32_t *, const int32_t *, int32_t, double *) {
    return fail(MSM_ERR_STATE, "msm_cost_triplet_batch: not implemented yet");
}
int msm_cost_triplet_octets(msm_cost *, const int32_t *, int32_t, double *) { return fail(MSM_ERR_STATE, "msm_cost_triplet_octets: not implemented yet"); }
int msm_cost_pairwise_batch(msm_cost *, const int32_t *, const int32_t *, const int32_t *, int32_t, double *) {
    return fail(MSM_ERR_STATE, "msm_cost_pairwise_batch: not implemented yet");
}
int msm_cost_pairwise_table(msm_cost *, double *) { return fail(MSM_ERR_STATE, "msm_cost_pairwise_table: not implemented yet"); }
int msm_cost_total(msm_cost *, const int32_t *, double *, double *) { return fail(MSM_ERR_STATE, "msm_cost_total: not implemented yet"); }
}
