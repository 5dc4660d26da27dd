// kernels.hpp -- launch wrappers implemented in kernels.hip (device pointers only).
#pragma once

#include "internal.hpp"

namespace msm {

// per-leaf sub-cell masks (see FlatOctree::node); writes 64 words per mask block
int launch_build_masks(msm_ctx *ctx, const DevTree &T, const double4 *d_nodebox, unsigned long long *d_mask);
int launch_query(msm_ctx *ctx, const DevTree &T, const double *d_q, int N, int *d_tri, int *d_vid, double *d_w, int mode);
int launch_closest_vertex(msm_ctx *ctx, const DevTree &T, const double *d_q, int N, int *d_out);
int launch_range(msm_ctx *ctx, const double *d_cp, int Ncp, const double *d_src, int Nsrc, const double *d_maxsep, double range, int cap,
                 uint32_t *d_slots, int *d_counts);

// rnl[(node*L + l)*9..] = estimate_rotation_matrix(cp[node], rot[node]*labels[l]); moved (optional, N x L x 3 AoS) = rot[node]*labels[l]
int launch_label_rotations(msm_ctx *ctx, const double *d_cp, int N, const double *d_rot, const double *d_labels, int L, double *d_rnl,
                           double *d_moved);

struct UnaryLaunch {
    DevTree tree;
    const double *tfeat;   // target features, V x D vertex-major
    int D;
    int N, L;
    const double *cp;      // 3 x N
    const double *rnl;     // N x L x 9 from launch_label_rotations
    const double *labels;  // 3 x L
    const double *src;     // 3 x Nsrc
    int Nsrc;
    const double *sfeat;   // D x Nsrc
    const double *cfw;     // rows x Nsrc or nullptr
    int cfw_rows;
    const int *pptr, *pidx;
    const double *absw;
    int pmax;
    int simmeasure;
    double *U;             // L x N
    unsigned long long *nsamples;
    // scratch owned by the cost object
    int ntri;                      // triangles in the target mesh
    double *tval;                  // one double per point sample (L * total patch points)
    unsigned long long *fix_list;  // capacity fix_cap
    unsigned int *fix_count;       // 2 words: [0] fix-up entries, [1] nodes to re-reduce
    unsigned int fix_cap;
    int *redo_list;                // N ints
};
int launch_unary_univariate(msm_ctx *ctx, const UnaryLaunch &u);

}  // namespace msm
