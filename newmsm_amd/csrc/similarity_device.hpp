// similarity_device.hpp -- per-point feature-vector similarity used by the multivariate cost functions
// (MultivariateNonLinearSRegDiscreteCostFunction::computeUnaryCost, M/DiscreteCostFunction.cpp:444-458, and the HO
// variant :601-618): sim(source features of one vertex, interpolated target features, per-dimension weights).
#pragma once

#include "geom.hpp"

namespace msm {

// target feature d of the sample = barycentric_interpolation of the three vertex rows f0,f1,f2 (vertex-major)
__device__ __forceinline__ double interp_feature(const double *f0, const double *f1, const double *f2, int d, double wa, double wb, double wc) {
    return wa * f0[d] + wb * f1[d] + wc * f2[d];
}

// serial over the D dimensions in the reference's order; the target values are recomputed in the second pass
// instead of being stored (3 loads + 5 flops each)
__device__ __forceinline__ double feature_vector_similarity(int sim, const double *sfeat, const double *cfw, int cfw_rows, int Nsrc, int sv, int D,
                                                            const double *f0, const double *f1, const double *f2, double wa, double wb, double wc) {
    auto W = [&](int d) { return (cfw && cfw_rows >= d + 1) ? cfw[(size_t)d * Nsrc + sv] : 1.0; };
    auto A = [&](int d) { return sfeat[(size_t)d * Nsrc + sv]; };
    if (sim == 2) {  // sparsesimkernel::corr, M/similarities.cpp:129-158
        double prod = 0.0, varA = 0.0, varB = 0.0, meanA = 0.0, meanB = 0.0, sum = 0.0;
        for (int d = 0; d < D; ++d) sum += W(d);
        for (int d = 0; d < D; ++d) {
            meanA += W(d) * A(d);
            meanB += W(d) * interp_feature(f0, f1, f2, d, wa, wb, wc);
        }
        if (sum > 0.0) {
            meanA /= sum;
            meanB /= sum;
        }
        for (int d = 0; d < D; ++d) {
            const double w = W(d), da = A(d) - meanA, db = interp_feature(f0, f1, f2, d, wa, wb, wc) - meanB;
            prod += w * da * db;
            varA += w * da * da;
            varB += w * db * db;
        }
        if (sum > 0.0) {
            prod /= sum;
            varA /= sum;
            varB /= sum;
        }
        const double r = (varA == 0.0 || varB == 0.0) ? 0.0 : prod / (sqrt(varA) * sqrt(varB));
        return 1 - (1 + r) * 0.5;  // get_sim_for_min, M/similarities.h:51-52
    }
    double prod = 0.0;  // sparsesimkernel::SSD, M/similarities.cpp:179-188
    for (int d = 0; d < D; ++d) {
        const double df = A(d) - interp_feature(f0, f1, f2, d, wa, wb, wc);
        prod += W(d) * df * df;
    }
    return sqrt(prod) / D;
}

}  // namespace msm
