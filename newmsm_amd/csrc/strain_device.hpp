// strain_device.hpp -- triangle strain energy (calculate_triangular_strain / triangle_strain / calculate_tri,
// M/reg_tools.cpp:267-313, :551-646, :698-743) as device functions shared by the pairwise-registration and the
// groupwise triplet costs.
#pragma once

#include "geom.hpp"

namespace msm {

__device__ __forceinline__ double det3(const double *M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// calculate_tri(const Point&), M/reg_tools.cpp:267-313: an orthonormal tangent pair for normal a
__device__ __forceinline__ void tangent_pair(const V3 &a, V3 &e1, V3 &e2) {
    V3 b = mk(1.0, 0.0, 0.0);
    V3 c = cross(a, b);
    double len = c.x * c.x + c.y * c.y + c.z * c.z;
    if (len == 0.0) {
        b = mk(0.0, 1.0, 0.0);
        c = cross(a, b);
        len = c.x * c.x + c.y * c.y + c.z * c.z;
    }
    len = sqrt(len);
    if (len == 0.0) len = 1;
    e1 = mk(c.x / len, c.y / len, c.z / len);
    b = cross(a, c);
    len = sqrt(b.x * b.x + b.y * b.y + b.z * b.z);
    if (len == 0) len = 1;
    e2 = mk(b.x / len, b.y / len, b.z / len);
}

// pow(x, e) of the strain energy and the regulariser exponent.  Every shipped configuration uses --k_exponent=2 / --regexp=2
// (or 1): x * x is the correctly rounded value of pow(x, 2), within an ulp of what any libm returns, and saves the
// ~150 dependent instructions of a general pow on the per-evaluation critical path.
__device__ __forceinline__ double pow_exp(double x, double e) { return e == 2.0 ? x * x : (e == 1.0 ? x : pow(x, e)); }

// the half of triangle_strain that depends on the ORIGINAL triangle only: the inverse of its 2-D edge matrix
struct StrainFrame {
    double i00, i01, i10, i11;
    bool dswap;  // the deformed triangle's tangent pair is swapped (the reference's second test of TRANS, M/reg_tools.cpp:721)
};

__device__ __forceinline__ double triangle_strain_from(const StrainFrame &fr, const double B[3][2], double mu, double kappa, double k_exp);

// triangle_strain, M/reg_tools.cpp:551-646 (strain energy density of the 2-D deformation gradient)
__device__ __forceinline__ double triangle_strain(const double A[3][2], const double B[3][2], double mu, double kappa, double k_exp) {
    const double c0 = A[1][0] - A[0][0], c1 = A[1][1] - A[0][1], c4 = A[2][0] - A[0][0], c5 = A[2][1] - A[0][1];
    const double det = c0 * c5 - c4 * c1;
    StrainFrame fr;
    fr.i00 = c5 / det, fr.i01 = -c4 / det, fr.i10 = -c1 / det, fr.i11 = c0 / det;
    fr.dswap = false;
    return triangle_strain_from(fr, B, mu, kappa, k_exp);
}

__device__ __forceinline__ double triangle_strain_from(const StrainFrame &fr, const double B[3][2], double mu, double kappa, double k_exp) {
    const double c0c = B[1][0] - B[0][0], c1c = B[1][1] - B[0][1], c4c = B[2][0] - B[0][0], c5c = B[2][1] - B[0][1];
    const double i00 = fr.i00, i01 = fr.i01, i10 = fr.i10, i11 = fr.i11;
    const double F00 = c0c * i00 + c4c * i10, F01 = c0c * i01 + c4c * i11;
    const double F10 = c1c * i00 + c5c * i10, F11 = c1c * i01 + c5c * i11;
    const double G[9] = {F00 * F00 + F10 * F10, F00 * F01 + F10 * F11, 0, F01 * F00 + F11 * F10, F01 * F01 + F11 * F11, 0, 0, 0, 1};
    const double I1 = G[0] + G[4] + G[8];
    const double I3 = det3(G);
    const double J = sqrt(I3);
    const double I1st = (I1 - 1.0) / J;
    const double R = (I1st <= 2) ? 1.0 : 0.5 * (I1st + sqrt(I1st * I1st - 4));
    const double Rs = pow_exp(R, k_exp), Js = pow_exp(J, k_exp);
    return 0.5 * (mu * (Rs + 1.0 / Rs - 2) + kappa * (Js + 1.0 / Js - 2));
}

// calculate_triangular_strain(Triangle, Triangle, ...), M/reg_tools.cpp:698-743, in two halves: what the original triangle
// contributes (its tangent frame, the 2-D coordinates of its vertices, the inverse edge matrix) ...
__device__ __forceinline__ StrainFrame strain_frame(const V3 o[3]) {
    const V3 nO = tri_normal(o[0], o[1], o[2]);
    V3 e1, e2;
    tangent_pair(nO, e1, e2);
    double TR[9] = {e1.x, e2.x, nO.x, e1.y, e2.y, nO.y, e1.z, e2.z, nO.z};
    V3 c1 = e1, c2 = e2;
    if (det3(TR) < 0) {  // swap the first two columns
        c1 = e2;
        c2 = e1;
        const double TS[9] = {e2.x, e1.x, nO.x, e2.y, e1.y, nO.y, e2.z, e1.z, nO.z};
        for (int k = 0; k < 9; ++k) TR[k] = TS[k];
    }
    double A[3][2];
    for (int i = 0; i < 3; ++i) {
        A[i][0] = o[i].x * c1.x + o[i].y * c1.y + o[i].z * c1.z;
        A[i][1] = o[i].x * c2.x + o[i].y * c2.y + o[i].z * c2.z;
    }
    const double c0 = A[1][0] - A[0][0], c1a = A[1][1] - A[0][1], c4 = A[2][0] - A[0][0], c5 = A[2][1] - A[0][1];
    const double det = c0 * c5 - c4 * c1a;
    StrainFrame fr;
    fr.i00 = c5 / det, fr.i01 = -c4 / det, fr.i10 = -c1a / det, fr.i11 = c0 / det;
    fr.dswap = det3(TR) < 0;  // the reference re-tests TRANS here, not TRANS2 (:721): kept as is
    return fr;
}
// ... and the deformed triangle against that frame
__device__ __forceinline__ double triangular_strain_from(const StrainFrame &fr, const V3 f[3], double mu, double kappa, double k_exp) {
    const V3 nF = tri_normal(f[0], f[1], f[2]);
    V3 t1, t2;
    tangent_pair(nF, t1, t2);
    const V3 d1 = fr.dswap ? t2 : t1, d2 = fr.dswap ? t1 : t2;
    double B2[3][2];
    for (int i = 0; i < 3; ++i) {
        B2[i][0] = f[i].x * d1.x + f[i].y * d1.y + f[i].z * d1.z;
        B2[i][1] = f[i].x * d2.x + f[i].y * d2.y + f[i].z * d2.z;
    }
    return triangle_strain_from(fr, B2, mu, kappa, k_exp);
}
__device__ __forceinline__ double triangular_strain(const V3 o[3], const V3 f[3], double mu, double kappa, double k_exp) {
    return triangular_strain_from(strain_frame(o), f, mu, kappa, k_exp);
}


}  // namespace msm
