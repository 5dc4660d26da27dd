// cost_internal.hpp -- the msm_cost handle shared by cost.cpp (unary) and cost_cliques.cpp (pairwise / triplet).
#pragma once

#include <vector>

#include "devbuf.hpp"
#include "kernels.hpp"

using msm::DevBuf;

struct msm_cost {
    msm_ctx *ctx = nullptr;
    msm_cost_params p{};
    msm_mesh *target = nullptr, *source = nullptr, *cpgrid = nullptr;
    std::vector<double> orig_xyz;  // _ORIG: source coordinates at set_meshes (3 x Nsrc)
    std::vector<double> ocp_xyz;   // _oCPgrid
    int D = 0;
    std::vector<double> sfeat;
    DevBuf<double> d_sfeat;
    DevBuf<double> d_sfeat_vm, d_cfw_vm;  // vertex-major copies for the multivariate reduction
    bool vm_valid = false;
    std::vector<double> cfw;
    int cfw_rows = 0;
    DevBuf<double> d_cfw;
    std::vector<double> maxsep;
    double mvdmax = 0;
    DevBuf<double> d_maxsep;
    int L = 0;
    std::vector<double> labels, rot;
    DevBuf<double> d_labels, d_rot;
    DevBuf<double> d_rnl, d_moved;  // per (node,label): rotation matrix and moved control point
    std::vector<int32_t> triplets, pairs;
    DevBuf<int32_t> d_triplets, d_pairs;
    DevBuf<double> d_orig, d_ocp;
    // control grid connectivity for the pairwise fold test
    DevBuf<int32_t> d_cp_tri, d_cp_tid_ptr, d_cp_tid;
    bool cp_conn_valid = false;
    DevBuf<int32_t> d_labeling;
    DevBuf<double> d_clique_out;
    DevBuf<double> d_ho_vals, d_ho_big;
    DevBuf<unsigned> d_ho_pending, d_ho_count;
    // fused fusion move of the HO classes (move_kernels.hip): per bin slot data prepared once per get_source_data()
    DevBuf<int32_t> d_slot_tri;
    DevBuf<int4> d_blk;
    DevBuf<double> d_slot_w, d_slot_sf, d_slot_cw, d_slot_wda, d_tri_frame, d_tri_stat;
    int64_t move_tails = 0;  // moves that needed the tail kernel
    // msm_cost_triplet_octets_prefetch: a label step queued ahead of its msm_cost_triplet_octets call (the optimiser's host-side solve of the previous
    // step runs meanwhile); taken by the call that asks for exactly this (labeling, label, E), dropped by any other entry point of this cost function
    struct PendingMove {
        bool valid = false;
        int kind = 0;  // 1: fused triclique move (k_ho_move), 2: strain-only packed move
        int label = 0;
        double *E = nullptr;
        uint64_t epoch = 0;  // msm_ctx::epoch when it was queued
        std::vector<int32_t> labeling;
        msm::CliqueArgs a;
        msm::MoveArgs m;
        msm::MoveLabels lab;
    } pending;
    int64_t prefetch_hits = 0, prefetch_drops = 0;
    DevBuf<unsigned> d_defer_list, d_defer_cnt;
    int move_nblk = 0, move_cap = 0, move_parity = 0;
    bool move_valid = false;
    // get_source_data products
    bool have_source = false;
    int ngroups = 0, pmax = 0;
    std::vector<int32_t> pptr, pidx;
    DevBuf<int32_t> d_pptr, d_pidx;
    std::vector<int32_t> order;
    DevBuf<int32_t> d_order;  // launch order of the groups (see msm_cost_get_source_data)
    DevBuf<uint32_t> d_fix_off;  // segment offsets of the fix-up list for (patches, L)
    bool fix_off_valid = false;
    std::vector<double> absw;
    DevBuf<int32_t> d_pidx_asc;  // patch members in ascending id (the host order); d_pidx holds the device order
    DevBuf<uint32_t> d_code;     // Morton codes of the source vertices
    DevBuf<double4> d_chunkb;    // k_range: bounding balls of the source vertices, 64 ids at a time
    bool pidx_asc_on_device = false;
    int patch_cap_hint = 0;
    DevBuf<double> d_absw, d_maxw;  // resample_weights: per control point / its input per source vertex
    // unary table
    DevBuf<double> d_U;
    bool table_valid = false;
    bool rotations_valid = false;
    std::vector<double> h_U;
    DevBuf<unsigned long long> d_counters;
    int64_t counters[4] = {0, 0, 0, 0};
    // scratch for the range kernel
    DevBuf<uint32_t> d_slots;
    // scratch of the unary kernels
    DevBuf<double> d_tval;
    DevBuf<int> d_stri;
    DevBuf<double> d_sw3;
    DevBuf<unsigned long long> d_fix_list;
    DevBuf<double> d_fix_pt;  // rotated points of the listed samples
    DevBuf<unsigned int> d_fix_count;
    // anatomical regularisation (regularisermode 4/5)
    msm_mesh *asphere = nullptr;
    int aVs = 0, aTs = 0;
    DevBuf<double> d_atarget, d_asrc, d_aw_val;
    DevBuf<int32_t> d_asrc_tri, d_aw_ptr, d_aw_cp, d_af_ptr, d_af_idx;
    bool have_anat = false;
    // optional event timing of the samples kernel
    bool timing = false;
    std::vector<hipEvent_t> ev0, ev1;
    int ev_next = 0, ev_count = 0;
    DevBuf<int> d_queues;  // nodes whose reduction waits for the fix-up kernel
    DevBuf<int> d_counts;
};


namespace msm {
int adaptive_weights(msm_mesh *in_mesh, msm_mesh *new_mesh, const double *excl, std::vector<int32_t> &row_ptr, std::vector<int32_t> &col,
                     std::vector<double> &val);
int query_host(msm_mesh *target, const double *q, int N, int *tri_id, int *vid, double *w, int mode, const char *what, const double *q_on_device = nullptr);
const Adjacency &mesh_adjacency(msm_mesh *m);
// (re)computes the per (control point, label) rotation matrices and moved control points if stale
int ensure_label_rotations(msm_cost *c);
int drop_pending_move(msm_cost *c);  // cost_cliques.cpp: waits for and discards a label step queued by msm_cost_triplet_octets_prefetch that nobody took
int ensure_vertex_major(msm_cost *c);  // d_sfeat_vm / d_cfw_vm
inline bool cost_is_ho(const msm_cost *c) { return c->p.kind == MSM_COST_HO_UNIVARIATE || c->p.kind == MSM_COST_HO_MULTIVARIATE; }
}  // namespace msm
