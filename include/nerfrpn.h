/*
 * nerfrpn.h -- C ABI of libnerfrpn_hip.so: the MI355X (gfx950) kernels of the 3D RPN-over-NeRF hot path.
 *
 * Contract (all entry points):
 *   - extern "C", plain pointers and sizes, no torch/ATen types.  Every pointer is a DEVICE pointer
 *     unless the parameter name starts with `h_` (host).  `stream` is a hipStream_t passed as void*.
 *   - Returns 0 on success, a negative nrpn_status otherwise; the message is available from
 *     nrpn_last_error() (thread-local).  Never aborts the process, never allocates device memory:
 *     outputs and workspaces are supplied by the caller (sizes from the *_workspace_bytes queries).
 *   - Asynchronous: kernels are enqueued on `stream`; no host synchronisation inside.
 *   - Activations are channels-last  [N][X][Y][Z][C]  (the reference's (W,L,H,C) on-disk order,
 *     reference datasets.py:55-56); C is the fastest dimension.  dtype codes: NRPN_F32, NRPN_BF16.
 *
 * Each group cites the reference interface it replaces (paths relative to /root/reference/nerf_rpn).
 * The reference has exactly one native op on this path (sort_vertices); everything else replaces
 * chains of torch ops / Python loops (SURVEY.md section 8a rows in brackets).
 */
#ifndef NERFRPN_H
#define NERFRPN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *nrpn_stream_t;

enum nrpn_status {
  NRPN_OK = 0,
  NRPN_ERR_ARG = -1,     /* bad shape / null pointer / unsupported combination */
  NRPN_ERR_LAUNCH = -2,  /* hipLaunch / hipGetLastError reported a failure     */
  NRPN_ERR_DEVICE = -3   /* wrong architecture / no device                      */
};

enum nrpn_dtype { NRPN_F32 = 0, NRPN_BF16 = 1 };

const char *nrpn_last_error(void);
int nrpn_abi_version(void);
/* 0 if device `ordinal` is a gfx950 part, NRPN_ERR_DEVICE otherwise. */
int nrpn_check_device(int ordinal);

/* ------------------------------------------------------------------------------------------------
 * Rotated-IoU family.  [a16, a17, a15]
 * nrpn_sort_vertices_f32 is the drop-in for the reference's pybind op
 *   sort_vertices.sort_vertices_forward(vertices f32[B,N,M,2], mask bool[B,N,M], num_valid i32[B,N]) -> i32[B,N,9]
 *   (model/rotated_iou/cuda_op/sort_vert.cpp:6-33, kernel sort_vert_kernel.cu:42-134); bn = B*N, m = M (24).
 * The *_iou3d_* entry points fuse corners -> edge intersections -> containment -> sort -> shoelace -> z-overlap
 *   (model/rotated_iou/oriented_iou_loss.py:6-107, box_intersection_2d.py:11-176) in registers, one lane per pair.
 * ---------------------------------------------------------------------------------------------- */
int nrpn_sort_vertices_f32(const float *vertices, const uint8_t *mask, const int32_t *num_valid,
                           int32_t *idx, int64_t bn, int m, nrpn_stream_t stream);
/* paired: b1,b2 [n,7] (x,y,z,w,h,d,theta) -> iou [n]            (cal_iou_3d, oriented_iou_loss.py:82-107) */
int nrpn_iou3d_obb_pair_f32(const float *b1, const float *b2, float *iou, int64_t n, nrpn_stream_t stream);
/* Differentiable IoU-type regression losses of RotatedIOULoss (model/rpn.py:133-164, fcos/loss.py:137-173) on paired boxes:
 *   pred, target [n,7] -> loss [n], grad [n,7] = d loss / d pred (target is a constant), iou [n] (optional).
 *   mode 0 'iou': -log((iou*union+1)/(union+1)); 1 'linear_iou': 1 - that ratio; 2 'giou' (cal_giou_3d, oriented_iou_loss.py:109-127);
 *   3 'diou' (cal_diou_3d, :129-148), enclosing box = smallest_bounding_box (min_enclosing_box.py:54-125).
 * Forward and gradient come out of ONE launch: the reference's arithmetic is evaluated on dual numbers (value + 7 partials), masks /
 * vertex sort / arg-min selections on the values, which is what autograd differentiates in the reference's 40-kernel chain. */
int nrpn_rotated_iou_loss_f32(const float *pred, const float *target, int64_t n, int mode, float *loss, float *grad, float *iou,
                              nrpn_stream_t stream);
/* 2-D projection smooth-L1 of the RPN, VALUE only (model/rpn.py:37-102 get_w2cs / project / obb2points_3d, 421-453): pred, target [n,box_dim]
 *   (6: AABB corners, 7: OBB), views [4][4][4] world->camera, intrinsics [3][3] -> out[0] = sum smooth_l1(uv_pred - uv_target; beta) / n /
 *   max_mesh_dim over 2 extreme points x 4 views x (u,v).  One workgroup, fixed summation order. */
int nrpn_projection_loss_f32(const float *pred, const float *target, int64_t n, int box_dim, const float *views, const float *intrinsics,
                             float beta, float max_mesh_dim, float *out, nrpn_stream_t stream);
/* Proposal metrics on the device (eval.py:14-81, 319-395).  [a25 / f1]
 * nrpn_recall_match_f32: the greedy GT <-> proposal matching of evaluate_box_proposals_recall on one scene's IoU matrix
 *   overlaps [P][G] (MODIFIED in place: retired rows / columns become -1) -> covered[j] = overlap recorded in round j, j < min(P, G);
 *   ties resolve to the first maximum (torch.max), G <= 1024.
 * nrpn_ap_mark: VOC-AP bookkeeping over ALL detections of the evaluation set: order [n] = detection indices in descending score order,
 *   best_iou [n] / key [n] = each detection's best IoU and (scene, GT) key in [0, num_keys); tp[r] = 1 iff detection order[r] has
 *   best_iou > iou_thresh and is the highest-ranked such detection of its key (eval.py:352-371).  first_ws: int32 [num_keys] scratch. */
int nrpn_recall_match_f32(float *overlaps, int num_proposals, int num_gt, float *covered, nrpn_stream_t stream);
int nrpn_ap_mark(const int64_t *order, const float *best_iou, const int64_t *key, int64_t n, int64_t num_keys, float iou_thresh,
                 int32_t *first_ws, uint8_t *tp, nrpn_stream_t stream);
/* all pairs: a [n,w], b [m,w] -> iou [n,m]; w = 6 (AABB x1..z2) or 7 (OBB)   (box_iou_3d, model/utils.py:387-458) */
int nrpn_iou3d_matrix_f32(const float *a, const float *b, float *iou, int64_t n, int64_t m, int box_dim,
                          nrpn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Greedy 3D NMS, per-level ("batched").  [a14]   (nms / batched_nms, model/utils.py:215-265)
 *   boxes [n,box_dim] must already be ordered score-descending inside each level run; `levels` (int32 [n],
 *   non-decreasing) restricts suppression to equal levels (nullptr = one level).  A box j is suppressed by an
 *   earlier kept box i of the same level iff !(IoU(i,j) <= thr).  `d_count` (int32 device scalar, may be
 *   nullptr => n_max) gives the live prefix length so no host sync is needed; entries >= *d_count get keep=0.
 *   keep: uint8 [n_max].  workspace: nrpn_nms3d_workspace_bytes(n_max) bytes.
 * ---------------------------------------------------------------------------------------------- */
size_t nrpn_nms3d_workspace_bytes(int64_t n_max);
int nrpn_nms3d(const float *boxes, const int32_t *levels, const int32_t *d_count, int64_t n_max, int box_dim,
               float thr, uint8_t *keep, void *workspace, nrpn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Segmented top-k / sort of fp32 scores, order = (score descending, index ascending).  [a13]
 *   (RegionProposalNetwork._get_top_n_idx, model/rpn.py:292-301; argsort in nms, utils.py:217)
 *   For each of nseg segments [h_offsets[s], h_offsets[s+1]) writes min(k, len) winners to
 *   out_idx[s*k ..] (int32, index relative to the start of `scores`) and out_val[s*k ..]; unused
 *   tail slots get idx = -1, val = -inf.  k <= 16384.
 * ---------------------------------------------------------------------------------------------- */
int nrpn_segmented_topk_f32(const float *scores, const int64_t *h_offsets, int nseg, int k,
                            int32_t *out_idx, float *out_val, nrpn_stream_t stream);
/* Same contract and results; segments of >= 65536 scores are selected by up to 256 workgroups (histogram / collect / tie / sort
 * launches, integer atomics only) instead of one.  `workspace`: nrpn_segmented_topk_workspace_bytes(nseg, k) bytes of device memory
 * (contents irrelevant, overwritten). */
size_t nrpn_segmented_topk_workspace_bytes(int nseg, int k);
int nrpn_segmented_topk_f32_ws(const float *scores, const int64_t *h_offsets, int nseg, int k,
                               int32_t *out_idx, float *out_val, void *workspace, size_t workspace_bytes, nrpn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Anchors and box coders.  [a8, a10, a11, a12]
 * Anchor table (device, int32 words) describes the pyramid so anchors are computed from their flat
 * index, never stored (AnchorGenerator3D, model/anchor.py:98-174; order (x,y,z,a), stride = mesh//grid):
 *   tab[0]=L levels, tab[1]=A anchors/cell, then per level l (8 words at 2+8l): gx,gy,gz, sx,sy,sz, first_lo, first_hi
 *   (first = int64 flat index of the level's first anchor); base anchors (float [L][A][6], already rounded)
 *   follow at word 2+8L, bit-cast.  nrpn_anchor_table_words(L,A) gives the word count.
 * ---------------------------------------------------------------------------------------------- */
int64_t nrpn_anchor_table_words(int levels, int anchors_per_cell);
/* anchors [count,6] for flat indices sel[0..count) (int64; nullptr => 0..count-1). */
int nrpn_anchors_f32(const int32_t *table, const int64_t *sel, int64_t count, float *anchors, nrpn_stream_t stream);
/* decode: deltas rows are gathered with the same index as the anchors.  coder 0 = AABB (6 deltas -> 6,
 * AABB_coder.py:86-137), 1 = midpoint-offset (8 deltas -> 7, midpoint_offset_coder.py:160-222).
 * deltas [T, 6|8] row-major in (x,y,z,a) anchor order; sel int64 [count] or nullptr; out [count, 6|7]. */
int nrpn_decode_boxes_f32(const int32_t *table, const float *deltas, const int64_t *sel, int64_t count,
                          int coder, float *boxes, nrpn_stream_t stream);
/* encode (regression targets) for gathered pairs: gt [count, 6|7], anchors from sel -> deltas [count, 6|8]
 * (AABB_coder.py:7-56, midpoint_offset_coder.py:106-158). */
int nrpn_encode_boxes_f32(const int32_t *table, const float *gt, const int64_t *sel, int64_t count,
                          int coder, float *deltas, nrpn_stream_t stream);

/* the same coders on explicit (row, anchor) pairs: in [count, 6|7|8], anchors [count,6] -> out; encode != 0 selects
 * gt -> deltas, else deltas -> boxes (BaseBBoxCoder.encode_single / decode_single). */
int nrpn_coder_pairs_f32(const float *in, const float *anchors, int64_t count, int coder, int encode, float *out,
                         nrpn_stream_t stream);
/* RPN head rows [cells][ld] fp32 (columns [0,A) = logits, [A, A+A*dw) = deltas, rest padding) <-> the reference's
 * flattened (x,y,z,a) order: logits [cells*A], deltas [cells*A, dw]  (permute_and_flatten, model/rpn.py:20-27).
 * unflatten is the backward: d_head (f32|bf16 [cells][ld], padding zeroed) = scale2[0]*g_logits , scale2[1]*g_deltas. */
int nrpn_head_flatten_f32(const float *head, int64_t cells, int ld, int anchors_per_cell, int dw, float *logits,
                          float *deltas, nrpn_stream_t stream);
int nrpn_head_unflatten(const float *g_logits, const float *g_deltas, int64_t cells, int ld, int anchors_per_cell, int dw,
                        const float *scale2, void *d_head, int dtype, nrpn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Proposal filter (eval).  [a13]   (filter_proposals, model/rpn.py:303-370; utils.py:268-367)
 *   In: candidates in per-level score-descending order: boxes [n, box_dim], logits [n], levels int32 [n],
 *   cand_valid uint8 [n] (0 for empty top-k slots).  Applies sigmoid, clip-to-grid (AABB clamp / OBB
 *   centre test that DROPS BOXES ONLY -- reference quirk B3, reproduced unless fix_obb_clip != 0),
 *   remove_small_boxes(min_size), score >= score_thresh, and stably compacts survivors to the front of
 *   out_boxes / out_scores / out_levels; *d_count = survivors.  All outputs sized n (n <= 16384).
 *   workspace: nrpn_filter_workspace_bytes(n, box_dim).
 * ---------------------------------------------------------------------------------------------- */
size_t nrpn_filter_workspace_bytes(int64_t n, int box_dim);
int nrpn_filter_candidates_f32(const float *boxes, const float *logits, const int32_t *levels,
                               const uint8_t *cand_valid, int64_t n, int box_dim, const float *h_grid_size3,
                               float min_size, float score_thresh, int fix_obb_clip, float *out_boxes,
                               float *out_scores, int32_t *out_levels, int32_t *d_count, void *workspace,
                               nrpn_stream_t stream);
/* Final gather: among the first *d_count entries with keep!=0, order by (score desc, index asc), keep at most
 * post_top_n: out_boxes [post_top_n, box_dim], out_scores, out_levels (float, as the reference returns),
 * *d_out_count.  n <= 16384.  (batched_nms tail, utils.py:264-265; rpn.py:362-364) */
int nrpn_select_kept_f32(const float *boxes, const float *scores, const int32_t *levels, const uint8_t *keep,
                         const int32_t *d_count, int64_t n, int box_dim, int post_top_n, float *out_boxes,
                         float *out_scores, float *out_levels, int32_t *d_out_count, nrpn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Target assignment (train).  [a18]   (assign_targets_to_anchors, rpn.py:240-290; Matcher, utils.py:98-211)
 *   gt_aabb [G,6] (OBB ground truth already rectified with obb2hbb_3d, misc.py:85-93), anchors from the table
 *   (T = total anchors).  h_ori_size3 (host, 3 floats) or nullptr: anchors whose cell starts at or beyond
 *   ceil(ori/stride) (zero-padded region) get quality -1 and label -1 (anchor.py:124-152).
 *   Outputs: labels f32 [T] in {1, 0, -1}; matched int32 [T] = clamp(match index, 0).
 *   workspace: G floats (per-GT maxima), zero-initialised by the call.
 * ---------------------------------------------------------------------------------------------- */
int nrpn_match_anchors_f32(const int32_t *table, int64_t total_anchors, const float *gt_aabb, int num_gt,
                           float fg_thresh, float bg_thresh, const float *h_ori_size3, float *labels,
                           int32_t *matched, float *workspace, nrpn_stream_t stream);
/* obb2hbb_3d: [n,7] -> [n,6] */
int nrpn_obb_to_aabb_f32(const float *obb, float *aabb, int64_t n, nrpn_stream_t stream);

/* Balanced positive / negative sampler (BalancedPositiveNegativeSampler, model/utils.py:35-98): labels [total] f32 (>= 1 positive, 0 negative,
 *   < 0 ignored) -> out_pos [max_pos], out_neg [batch] int64 ascending + counts int32[3] = {num_pos, num_neg, error}, with
 *   num_pos = min(#pos, max_pos), num_neg = min(#neg, batch - num_pos) and each subset uniformly random: the k smallest
 *   (splitmix64(seed, index) >> 32, index) of the class.  No host read-back inside (the reference needs two torch.where sizes and
 *   two randperm sorts per scene); the result depends on (labels, seed) only.  workspace: nrpn_sample_workspace_bytes(), 8-byte aligned. */
size_t nrpn_sample_workspace_bytes(void);
int nrpn_sample_pos_neg(const float *labels, int64_t total, int max_pos, int batch, int64_t seed, void *workspace, int64_t *out_pos,
                        int64_t *out_neg, int32_t *counts, nrpn_stream_t stream);
/* ------------------------------------------------------------------------------------------------
 * Sampled RPN losses with fused backward.  [a20]  (compute_loss, rpn.py:372-419)
 *   logits [T], deltas [T,dw], targets [npos,dw] (already encoded for the sampled positives), pos/neg int64 index
 *   lists.  loss[0] = BCE-with-logits mean over pos+neg (labels 1/0); loss[1] = smooth-L1(beta, sum over pos) /
 *   (npos+nneg).  Writes d(loss0)/d(logits) into g_logits and d(loss1)/d(deltas) into g_deltas at the sampled rows
 *   (caller zero-fills the rest).
 * ---------------------------------------------------------------------------------------------- */
int nrpn_rpn_sampled_loss_f32(const float *logits, const float *deltas, int dw, const float *targets,
                              const int64_t *pos, int64_t npos, const int64_t *neg, int64_t nneg, float beta,
                              float *loss2, float *g_logits, float *g_deltas, nrpn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Conv3d family, channels-last, implicit GEMM on MFMA.  [a3, a4, a7, a21]
 *   (torch.nn.Conv3d inside VGG_FPN feature_extractor.py:331-358, FPN fpn.py:109-110, RPNHead anchor.py:190-198)
 *   x [N,X,Y,Z,Cin], y [N,X,Y,Z,Cout] (k3: stride 1, pad 1; k1: stride 1, pad 0), dtype f32 or bf16 (fp32 accumulate).
 *   Packed weights (nrpn_pack_conv_weight, from the reference layout [Cout][Cin][kx][ky][kz] fp32):
 *     forward form  wp_fwd  [taps][rows_total][Cin]          tap = (kx*3+ky)*3+kz
 *     dgrad form    wp_dgrad[taps][Cin][rows_total]          taps reversed, so dgrad IS nrpn_conv3d_fwd on dy with
 *                                                            (cin, cout, wrows) := (rows_total, Cin, Cin)
 *   rows_total >= Cout lets several reference convs share one GEMM (cls_logits + bbox_pred -> 128 rows, the
 *   remaining rows zero); `row_offset` places this weight's rows.  `wrows` below = rows_total of the packed buffer.
 *   flags: NRPN_CONV_BIAS (bias f32 [Cout]), NRPN_CONV_RELU, NRPN_CONV_OUT_F32 (bf16 inputs, fp32 output rows).
 *   Cin*elemsize must be a multiple of 64 bytes.
 * ---------------------------------------------------------------------------------------------- */
enum { NRPN_CONV_BIAS = 1, NRPN_CONV_RELU = 2, NRPN_CONV_OUT_F32 = 4,
       /* timing diagnosis of the 256x256 kernel only -- RESULTS ARE WRONG with these set (tools/bench_tile.py): every tap reads the
        * centre voxel ("ideal memory"), resp. the per-K-step barrier + DMA drain is skipped ("free-running waves") */
       NRPN_CONV_DEBUG_ALIAS_TAPS = 256, NRPN_CONV_DEBUG_NO_SYNC = 512,
       /* experiment (results stay correct): waves 4-7 of the 256x256 kernel issue their LDS-DMA two sub-steps later than waves 0-3 */
       NRPN_CONV_DEBUG_STAGGER = 1024,
       /* tools only (results stay correct): 2-bit A/B variant selector of the halo kernel (bits 12-13), see conv_halo.hip */
       NRPN_CONV_DEBUG_VARIANT = 0x3000 };
int nrpn_pack_conv_weight(const float *w_ref, int cout, int cin, int taps, int dtype, void *wp_fwd, void *wp_dgrad,
                          int rows_total, int row_offset, nrpn_stream_t stream);
/* packed fp32 partial weight gradients [slices][taps][rows_total][Cin] (output of nrpn_conv3d_wgrad) -> sum over the
 * slices in the reference layout [cout][Cin][taps] (optionally accumulating into gw_ref, e.g. a flat-arena slot) */
int nrpn_unpack_conv_wgrad(const float *gw_packed, int cout, int cin, int taps, int rows_total, int row_offset,
                           float *gw_ref, int accumulate, int slices, nrpn_stream_t stream);
/* `workspace` (nrpn_conv3d_fwd_workspace_bytes; may be NULL = never split): for small grids the K loop is split over
 * K slices whose fp32 partials are stored to it and summed in slice order (deterministic, no atomics) so the 10^3 / 5^3
 * pyramid levels still fill 256 CUs. */
size_t nrpn_conv3d_fwd_workspace_bytes(int n, int gx, int gy, int gz, int cin, int cout, int ksize, int dtype);
/* relu_mask (optional, [N*X*Y*Z][Cout] in x's dtype): outputs are zeroed where relu_mask <= 0 -- used by dgrad launches whose input
 * gradient feeds a fused conv+ReLU layer: the ReLU backward of that layer costs no extra pass (mask = that layer's activations). */
int nrpn_conv3d_fwd(const void *x, const void *wp, const float *bias, void *y, int n, int gx, int gy, int gz, int cin,
                    int cout, int wrows, int ksize, int dtype, int flags, void *workspace, const void *relu_mask,
                    nrpn_stream_t stream);
/* Per-call plan + fused-epilogue extras of a forward / dgrad launch.  Zero-initialise, set `size`, then the fields you need; a NULL
 * opts pointer (or all defaults) is exactly nrpn_conv3d_fwd.  Every kernel-selection decision of a call is a pure function of
 * (shape, opts, process defaults): concurrent calls with different plans share no mutable state.
 *   scale      f32 [Cout]: y = acc * scale + bias -- eval-mode BatchNorm3d folded into the conv that feeds it (feature_extractor.py:345-358:
 *              scale = gamma / sqrt(running_var + eps), bias = (conv bias - running_mean) * scale + beta), with NRPN_CONV_RELU on top
 *   relu_mask  as in nrpn_conv3d_fwd;  stats: as in nrpn_conv3d_fwd_stats (rows: nrpn_conv3d_fwd_stats_rows_ex with the same opts)
 *   tile       NRPN_TILE_*: 0 = chosen per shape; lds_dma / stagger / big_split: -1 = default, 0 / 1; kstep_bytes: 0 = default, 64, 128
 *   debug      tools only: NRPN_CONV_DEBUG_* bits (timing variants, wrong results) */
enum { NRPN_TILE_AUTO = 0, NRPN_TILE_128 = 128, NRPN_TILE_256X128_WS = 256, NRPN_TILE_256X256 = 512, NRPN_TILE_256X256_W4 = 1024,
       NRPN_TILE_HALO = 2048 /* 3x3x3 bf16 only: 4x8x8 voxel blocks whose input halo is staged once per channel chunk */ };
typedef struct nrpn_conv_opts {
  int32_t size;
  int32_t tile;
  int32_t lds_dma;
  int32_t kstep_bytes;
  int32_t stagger;
  int32_t big_split;
  int32_t debug;
  int32_t halo_pairing;   /* halo form only: 0 = default, 1 = pair taps across channel-chunk boundaries (Cin % 128 == 0), 2 = 14 K-steps per chunk */
  const float *scale;
  const void *relu_mask;
  float *stats;
} nrpn_conv_opts;
int nrpn_conv3d_fwd_ex(const void *x, const void *wp, const float *bias, void *y, int n, int gx, int gy, int gz, int cin,
                       int cout, int wrows, int ksize, int dtype, int flags, void *workspace, const nrpn_conv_opts *opts,
                       nrpn_stream_t stream);
size_t nrpn_conv3d_fwd_workspace_bytes_ex(int n, int gx, int gy, int gz, int cin, int cout, int ksize, int dtype,
                                          const nrpn_conv_opts *opts);
int nrpn_conv3d_fwd_plan_ex(int n, int gx, int gy, int gz, int cin, int cout, int ksize, int dtype, const nrpn_conv_opts *opts);
int nrpn_conv3d_fwd_stats_rows_ex(int n, int gx, int gy, int gz, int cin, int cout, int ksize, int dtype, const nrpn_conv_opts *opts);
/* wgrad: the voxel axis is cut into S = nrpn_conv3d_wgrad_slices(...) slices; every (tile, tap, slice) workgroup writes
 * its partial with plain stores into gw_packed f32 [S][taps][wrows][Cin] (fully overwritten: no memset, no atomics --
 * cross-XCD fp32 atomics were ~1/3 of the kernel time); nrpn_unpack_conv_wgrad sums the slices.
 * optional gbias f32 [Cout] = column sums of dy (per-slice partials in the workspace, summed in slice order: deterministic).
 * `workspace` (always required) = k3 tap masks + the bias partials. */
size_t nrpn_conv3d_wgrad_workspace_bytes(int n, int gx, int gy, int gz, int cin, int cout, int wrows, int ksize, int dtype);
int nrpn_conv3d_wgrad_slices(int n, int gx, int gy, int gz, int cin, int cout, int wrows, int ksize, int dtype);
/* accumulate_bias = flags: NRPN_WGRAD_ACC_BIAS: the column sums are ADDED to gbias (e.g. a slot of a flat gradient arena) instead of
 * overwriting; NRPN_WGRAD_MASK_READY: `workspace` still holds the tap masks an earlier call wrote for the same (n, gx, gy, gz) grid /
 * segment list, so they are not rebuilt (callers that keep one workspace per grid shape save a launch per layer);
 * NRPN_WGRAD_DEFER_BIAS: gbias must still be non-NULL to request the partial sums, but it is not written by this call. */
enum { NRPN_WGRAD_ACC_BIAS = 1, NRPN_WGRAD_MASK_READY = 2,
       NRPN_WGRAD_DEFER_BIAS = 4 /* leave the bias partials in the workspace: nrpn_reduce_slices finishes them */ };
int nrpn_conv3d_wgrad(const void *x, const void *dy, float *gw_packed, float *gbias, int n, int gx, int gy, int gz,
                      int cin, int cout, int wrows, int ksize, int dtype, int accumulate_bias, void *workspace,
                      nrpn_stream_t stream);
/* Ragged voxel lists: nseg <= 16 grids laid end to end in x / y / dy (dims: host int32 [nseg*3] = X,Y,Z of each segment), e.g. the
 * (level, scene) maps of a weight-sharing RPN / FCOS head, which then run as ONE launch per layer instead of one per level
 * (the 10^3 / 5^3 levels are launch- and latency-bound on their own).  Same semantics as the per-grid calls on every segment;
 * workspace sizes / slice counts: call the queries above with n = 1, gx = total voxels, gy = gz = 1. */
int nrpn_conv3d_fwd_ragged(const void *x, const void *wp, const float *bias, void *y, int nseg, const int32_t *dims, int cin,
                           int cout, int wrows, int ksize, int dtype, int flags, void *workspace, nrpn_stream_t stream);
int nrpn_conv3d_wgrad_ragged(const void *x, const void *dy, float *gw_packed, float *gbias, int nseg, const int32_t *dims,
                             int cin, int cout, int wrows, int ksize, int dtype, int accumulate_bias, void *workspace,
                             nrpn_stream_t stream);
/* ---- Row-list ("sampled cone") forms.  In training the RPN loss reads the head at the sampled anchors only (reference model/rpn.py:389-420;
 * "During training, boxes pred and scores are unused", rpn.py:506), so the head -- conv_depth x [Conv3d k3 + ReLU] + the 1x1x1 cls / bbox
 * convs of reference model/anchor.py:RPNHead, shared by all pyramid levels -- is needed only on the receptive-field cones of those voxels,
 * and its gradients are non-zero only there.  nrpn_cone_build turns the sampler's output into the sorted voxel lists S0 c S1 c ... c S_depth
 * (S_k = 3x3x3 dilation of S_{k-1} inside each grid) over the ragged voxel space of `dims` (segment = level * n_scenes + scene, host int32
 * [nlevels * n_scenes * 3]); the *_rows calls run a conv on exactly the listed rows.  List entry = two uint32 per row: { voxel id, in-bounds bits
 * of the 27 taps | segment << 27 }.
 *   pos / neg: device int64 [n_scenes][pos_stride | neg_stride] anchor indices in one scene's flat (level, x, y, z, a) order, as
 *   nrpn_sample_pos_neg writes them; counts: device int32 [3 * n_scenes] = (kp, kn, err) per scene (its out_counts);
 *   level_anchor_off: host int64 [nlevels + 1]; lists: device uint32 [depth + 1][cap][2], cap >= total voxels; counts_out: device int32
 *   [depth + 2] = |S_0| .. |S_depth|, then an error flag (an anchor index outside the pyramid).  No host synchronisation. */
size_t nrpn_cone_workspace_bytes(int64_t total_voxels);
int nrpn_cone_build(const int64_t *pos, const int64_t *neg, const int32_t *counts, int n_scenes, int64_t pos_stride, int64_t neg_stride,
                    int nlevels, const int64_t *level_anchor_off, int num_anchors, const int32_t *dims, int depth, uint32_t *lists,
                    int64_t cap, int32_t *counts_out, void *workspace, nrpn_stream_t stream);
/* forward / dgrad on the `nrows` listed output rows: x, y and relu_mask are [total voxels][C] tensors indexed by voxel id; rows outside the
 * list are not written.  128-row tiles; short lists run on K slices (fp32 partials in `workspace`, >= nrpn_conv3d_fwd_rows_workspace_bytes; NULL =
 * one slice) summed in slice order by a scatter epilogue.  Cin * elemsize must be a multiple of 128 bytes. */
size_t nrpn_conv3d_fwd_rows_workspace_bytes(int64_t nrows, int cin, int cout, int ksize, int dtype);
int nrpn_conv3d_fwd_rows(const void *x, const void *wp, const float *bias, void *y, const uint32_t *rows, int64_t nrows, int nseg,
                         const int32_t *dims, int cin, int cout, int wrows, int ksize, int dtype, int flags, const void *relu_mask,
                         void *workspace, nrpn_stream_t stream);
/* wgrad over the `nrows` listed voxels (the K extent): partial layout / slice count as nrpn_conv3d_wgrad with n = 1, gx = nrows, gy = gz = 1;
 * workspace: >= max(256, slices * wrows * 4) bytes (bias partials; NRPN_WGRAD_* flags as above, MASK_READY is implied). */
int nrpn_conv3d_wgrad_rows(const void *x, const void *dy, float *gw_packed, float *gbias, const uint32_t *rows, int64_t nrows, int nseg,
                           const int32_t *dims, int cin, int cout, int wrows, int ksize, int dtype, int accumulate_bias, void *workspace,
                           nrpn_stream_t stream);
/* kernel selection of a forward / dgrad launch: 0 = 128-row tile, 1 = 256x256 tile (8 waves), 2 = 256x256 tile on K slices, 3 = 128-row
 * tile on K slices, 4 = wave-specialised 256x128, 5 = 256x256 tile on 4 waves, 6 = the same on K slices,
 * 7 = halo form of the 3x3x3 kernel;  of a wgrad launch: 1 = 256x256 tile, 0 = 128x128 (tests assert coverage with these) */
int nrpn_conv3d_fwd_plan(int n, int gx, int gy, int gz, int cin, int cout, int ksize, int dtype);
/* BatchNorm statistics out of the conv epilogue (the conv -> BatchNorm3d pairs of feature_extractor.py:288-377 in training mode):
 * nrpn_conv3d_fwd_stats = nrpn_conv3d_fwd that also writes per-row-group partial (sum, sum of squares) of the STORED bf16 outputs into
 * stats f32 [P][2][Cout], P = nrpn_conv3d_fwd_stats_rows(...) (0 = this shape's kernel has no fused statistics: K-sliced / fp32 / narrow
 * K-step -- run nrpn_bn_stats on the output instead); nrpn_bn_stats_finalize turns the partials into mean / biased variance and updates
 * the running statistics exactly as nrpn_bn_stats does for its own slab partials (fp64 accumulation, fixed order). */
int nrpn_conv3d_fwd_stats_rows(int n, int gx, int gy, int gz, int cin, int cout, int ksize, int dtype);
int nrpn_conv3d_fwd_stats(const void *x, const void *wp, const float *bias, void *y, int n, int gx, int gy, int gz, int cin,
                          int cout, int wrows, int ksize, int dtype, int flags, float *stats, nrpn_stream_t stream);
int nrpn_bn_stats_finalize(const float *partials, int nparts, int64_t rows, int c, float *mean, float *var, float *running_mean,
                           float *running_var, float momentum, nrpn_stream_t stream);
int nrpn_conv3d_wgrad_plan(int n, int gx, int gy, int gz, int cin, int cout, int wrows, int ksize, int dtype);
/* State: the library keeps NO per-call or per-stream state -- every buffer, workspace and stream comes from the caller, and a caller that needs
 * a non-default kernel plan passes nrpn_conv_opts to the *_ex entry points.  What is process-wide: (a) a (kernel, device) cache of granted
 * dynamic-LDS limits (mutex-protected, idempotent); (b) the thread-local message behind nrpn_last_error(); (c) the measurement switches of
 * include/nerfrpn_tools.h (process defaults for A/B timing runs: tools and tests only -- they are NOT part of this boundary and the product
 * path never calls them). */
/* Stem: Conv3d(4 -> Cout, k7, pad 3, stride 1|2) on [N,X,Y,Z,4] (feature_extractor.py:336,341) as an im2col GEMM
 * whose A operand is gathered tap by tap.  Packed stem weights: [Cout][Kpad], k = tap*4 + c, Kpad = nrpn_stem_kpad(dtype).
 * Output grid: (X - 1)/stride + 1 per axis. */
int nrpn_stem_kpad(int dtype);
int nrpn_pack_stem_weight(const float *w_ref, int cout, int dtype, void *wp, nrpn_stream_t stream);
/* stem wgrad writes S = nrpn_stem_wgrad_slices(...) partials [S][Cout][Kpad]; the unpack sums them */
int nrpn_stem_wgrad_slices(int n, int gx, int gy, int gz, int cout, int stride, int dtype);
/* floats of one slice partial: [Cout][Kpad], or -- stride 2, even Z, Cout 64: the "z-row" kernel, 49 GEMMs over contiguous 8-z x 4-c
 * input runs instead of an im2col gather -- [49][Cout][32]; pass it to the unpack so it knows the layout */
int64_t nrpn_stem_wgrad_slice_floats(int n, int gx, int gy, int gz, int cout, int stride, int dtype);
int nrpn_unpack_stem_wgrad(const float *gw_packed, int cout, int dtype, float *gw_ref, int accumulate, int slices,
                           int64_t slice_floats, nrpn_stream_t stream);
int nrpn_conv3d_stem_fwd(const void *x, const void *wp, const float *bias, void *y, int n, int gx, int gy, int gz,
                         int cout, int stride, int dtype, int flags, nrpn_stream_t stream);
/* "Halo" form of the stem forward for bf16, stride 2, Cout 64 and an even Z (nrpn_stem_halo_supported): a workgroup stages the input halo
 * of a 4x4x16 block of output voxels in LDS once and reads every MFMA A fragment from it (the im2col form above moves 2.7 kB per output
 * voxel through the L1 -> LDS path).  Weights: nrpn_pack_stem_weight_halo -> bf16 [Cout][nrpn_stem_halo_kpad()].  Epilogue:
 * acc * scale + bias (scale optional: eval-mode BatchNorm fold), ReLU with NRPN_CONV_RELU. */
int nrpn_stem_halo_supported(int gz, int cout, int stride, int dtype);
int nrpn_stem_halo_kpad(void);
int nrpn_pack_stem_weight_halo(const float *w_ref, int cout, void *wp, nrpn_stream_t stream);
int nrpn_conv3d_stem_fwd_halo(const void *x, const void *wp, const float *bias, const float *scale, void *y, int n, int gx, int gy,
                              int gz, int cout, int flags, nrpn_stream_t stream);
/* the same with the nrpn_conv_opts extras that apply to the stem: `scale` (eval-mode BatchNorm folded in) with NRPN_CONV_RELU on top */
int nrpn_conv3d_stem_fwd_ex(const void *x, const void *wp, const float *bias, void *y, int n, int gx, int gy, int gz,
                            int cout, int stride, int dtype, int flags, const nrpn_conv_opts *opts, nrpn_stream_t stream);
size_t nrpn_stem_wgrad_workspace_bytes(int n, int gx, int gy, int gz, int cout, int stride, int dtype);
int nrpn_conv3d_stem_wgrad(const void *x, const void *dy, float *gw_packed, float *gbias, int n, int gx, int gy, int gz,
                           int cout, int stride, int dtype, int accumulate_bias, void *workspace, nrpn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm3d / ReLU / MaxPool3d / nearest-upsample-add, channels-last.  [a3, a4, a21]
 *   (feature_extractor.py:337-358, fpn.py:150-155)
 * ---------------------------------------------------------------------------------------------- */
/* per-channel batch statistics over rows = N*X*Y*Z: mean, biased var (f32 [C]); updates running stats
 * (momentum m, unbiased var) when running_mean != nullptr.  workspace: nrpn_bn_workspace_bytes(rows, c). */
size_t nrpn_bn_workspace_bytes(int64_t rows, int c);
int nrpn_bn_stats(const void *x, int64_t rows, int c, int dtype, float *mean, float *var, float *running_mean,
                  float *running_var, float momentum, void *workspace, nrpn_stream_t stream);
/* y = relu?((x - mean) * rsqrt(var + eps) * gamma + beta) */
int nrpn_bn_apply(const void *x, void *y, int64_t rows, int c, int dtype, const float *mean, const float *var,
                  const float *gamma, const float *beta, float eps, int relu, nrpn_stream_t stream);
/* backward of bn_apply(+relu) in train mode: given x (conv output), dy and either beta (the ReLU mask is then recomputed
 * from x with the bn_apply expression and y is not read -- one tensor less of HBM traffic) or y (post-activation),
 * writes dx and dgamma/dbeta (f32 [C], overwritten); acc_dgamma / acc_dbeta (optional) are additionally incremented. */
int nrpn_bn_backward(const void *x, const void *y, const void *dy, void *dx, int64_t rows, int c, int dtype,
                     const float *mean, const float *var, const float *gamma, const float *beta, float eps, int relu,
                     float *dgamma, float *dbeta, float *acc_dgamma, float *acc_dbeta, void *workspace, nrpn_stream_t stream);
int nrpn_relu_backward(const void *y, const void *dy, void *dx, int64_t count, int dtype, nrpn_stream_t stream);
/* MaxPool3d(k, stride s, pad p, ceil_mode) forward writes int8 argmax offsets (window-local) for the backward. */
int nrpn_pool_out_size(int in, int k, int s, int p, int ceil_mode);
int nrpn_maxpool3d_fwd(const void *x, void *y, int8_t *argmax, int n, int gx, int gy, int gz, int c, int k, int s,
                       int p, int ceil_mode, int dtype, nrpn_stream_t stream);
int nrpn_maxpool3d_bwd(const void *dy, const int8_t *argmax, void *dx, int n, int gx, int gy, int gz, int c, int k,
                       int s, int p, int ceil_mode, int dtype, nrpn_stream_t stream);
/* fine += nearest_upsample(coarse) (legacy floor(dst*in/out) index rule, fpn.py:153-155); backward reduces. */
int nrpn_upsample_add_fwd(void *fine, const void *coarse, int n, int fx, int fy, int fz, int cx, int cy, int cz, int c,
                          int dtype, nrpn_stream_t stream);
int nrpn_upsample_add_bwd(const void *dfine, void *dcoarse, int n, int fx, int fy, int fz, int cx, int cy, int cz,
                          int c, int dtype, int accumulate, nrpn_stream_t stream);
/* ResNet bottleneck pieces (feature_extractor.py:31-68): strided 1x1x1 conv = voxel subsample (x*s, y*s, z*s) + 1x1x1 GEMM;
 * backward != 0 scatters a [N,ceil(X/s),..,C] gradient back into a zero-filled [N,X,Y,Z,C].  y = relu?(a + b) is the
 * residual join (its backward is nrpn_relu_backward on y). */
int nrpn_subsample3d(const void *src, void *dst, int n, int gx, int gy, int gz, int c, int stride, int backward, int dtype,
                     nrpn_stream_t stream);
int nrpn_add_relu(const void *a, const void *b, void *y, int64_t count, int relu, int dtype, nrpn_stream_t stream);
/* Scene ingest [a1] (datasets.py:39-63,165-167; ScanNet :227-231): src = the on-disk (W,L,H,4) rgb-sigma array (f32, or uint8
 * -> /255) on the device; dst = channels-last [W,L,H,4] in the compute dtype; alpha_mode 0 none, 1 density_to_alpha
 * clip(1-exp(-exp(s)/100),0,1), 2 the ScanNet variant clip(1-exp(-max(s,0)/100),0,1) on channel 3.  One pass instead of
 * numpy alpha + host transpose + H2D of fp32 + device transpose. */
int nrpn_ingest_rgbsigma(const void *src, int src_is_u8, void *dst, int64_t voxels, int alpha_mode, int dtype, nrpn_stream_t stream);
/* Ingest + training augmentation in one pass (datasets.py:109-163, 291-329): rot90 (z_up: transpose(x,y)+flip x, else transpose(x,z)+
 * flip z), flips of axis 0 and axis 1 (z_up) / 2, then -- if h_xform (host, 9 floats = R(angle)*scale row-major) is not NULL --
 * rotate_and_scale_scene's trilinear resample (grid_sample align_corners=True, zero padding) of the alpha-converted scene.
 * src (W,L,H,4) f32|u8 -> dst [OW,OL,OH,4] dtype, (OW,OL,OH) = (L,W,H) / (H,L,W) when rotated (z_up / not), else (W,L,H). */
int nrpn_ingest_augment(const void *src, int src_is_u8, void *dst, int w, int l, int h, int alpha_mode, int dtype, int rot90,
                        int z_up, int flip0, int flip1, const float *h_xform, nrpn_stream_t stream);
/* layout / dtype conversion between the reference's [N,C,X,Y,Z] f32 and channels-last f32|bf16 */
int nrpn_ncdhw_to_ndhwc(const float *src, void *dst, int n, int c, int64_t voxels, int dtype, nrpn_stream_t stream);
int nrpn_ndhwc_to_ncdhw(const void *src, float *dst, int n, int c, int64_t voxels, int dtype, nrpn_stream_t stream);
int nrpn_cast(const void *src, void *dst, int64_t count, int src_dtype, int dst_dtype, nrpn_stream_t stream);
/* Column sums of f32 x [rows][C] -> out [C] (+= when accumulate): the bias gradient of a convolution (sum of dy over the voxels; torch's
 * Conv3d backward, feature_extractor.py:345-358) in the bf16x3 mode, where the weight-gradient launch sees split planes.  Deterministic
 * (fixed slab order, fp64 finish).  workspace: nrpn_column_sum_workspace_bytes. */
size_t nrpn_column_sum_workspace_bytes(int64_t rows, int c);
int nrpn_column_sum_f32(const float *x, int64_t rows, int c, float *out, int accumulate, void *workspace, nrpn_stream_t stream);
/* bf16x3 operands of the parity-grade fast mode: src f32 [rows][C] -> hi = bf16(x), lo = bf16(x - hi), written as an interleaved operand
 * bf16 [rows][3C] (forward / dgrad: the K axis tripled; segment s holds lo when bit s of ipattern is set, else hi -- activations 0b100,
 * weights 0b010: x*w ~ hi*whi + hi*wlo + lo*whi) and / or as nplanes planes bf16 [nplanes][rows][C] stacked on the batch axis (weight
 * gradient: x planes 0b010, dy planes 0b100).  Either destination may be NULL.  C % 8 == 0.  Replaces nothing in the reference: it lets
 * the bf16 MFMA kernels reproduce torch.nn.Conv3d's fp32 arithmetic (feature_extractor.py:345-358) to fp32 accumulation error. */
int nrpn_split_bf16x3(const float *src, int64_t rows, int c, void *interleaved, int ipattern, void *planes, int nplanes, int ppattern,
                      nrpn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Swin-3D backbone pieces, channels-last tokens [N,X,Y,Z,C].  [a6]  (feature_extractor.py:382-789)
 *   Every token-wise Linear is nrpn_conv3d_fwd with ksize 1; window partition / cyclic shift / padding are index
 *   arithmetic inside the attention kernels.
 * ---------------------------------------------------------------------------------------------- */
/* patch_partition gather (feature_extractor.py:700-710): [N,X,Y,Z,4] -> [N,X/p,Y/p,Z/p,4p^3], inner order (c,dx,dy,dz)
 * = the flattened Conv3d(4,E,k=p,s=p) weight, so the patch embedding is a 1x1x1 GEMM on weight.view(E, 4p^3). */
int nrpn_patchify(const void *x, void *y, int n, int gx, int gy, int gz, int patch, int dtype, nrpn_stream_t stream);
/* nn.LayerNorm(C) over rows tokens; mean / rstd (f32 [rows]) are saved for the backward, which overwrites dgamma/dbeta
 * (block partials in `workspace` = nrpn_layernorm_workspace_bytes(rows, c), summed by a second kernel: no atomics). */
int nrpn_layernorm_fwd(const void *x, void *y, const float *gamma, const float *beta, float *mean, float *rstd, int64_t rows,
                       int c, float eps, int dtype, nrpn_stream_t stream);
size_t nrpn_layernorm_workspace_bytes(int64_t rows, int c);
int nrpn_layernorm_bwd(const void *x, const void *dy, void *dx, const float *gamma, const float *mean, const float *rstd,
                       float *dgamma, float *dbeta, int64_t rows, int c, int dtype, int accumulate_params, void *workspace,
                       nrpn_stream_t stream);   /* accumulate_params != 0: dgamma / dbeta += (e.g. slots of a flat gradient arena) instead of = */
/* exact (erf) GELU; backward != 0: out = dy * gelu'(x) */
int nrpn_gelu(const void *x, const void *dy, void *out, int64_t count, int backward, int dtype, nrpn_stream_t stream);
/* y = (a ? a : 0) + scale[n] * b : residual join with the StochasticDepth("row") factor (scale == NULL: 1) */
int nrpn_scale_add(const void *a, const void *b, const float *scale, void *y, int n, int64_t per_sample, int dtype,
                   nrpn_stream_t stream);
/* PatchMerging gather (feature_extractor.py:396-421): [N,X,Y,Z,C] -> [N,ceil(X/2),ceil(Y/2),ceil(Z/2),8C] (odd sizes zero
 * padded), block order x0..x7 of the reference; backward != 0 maps a [.., 8C] gradient back to [N,X,Y,Z,C]. */
int nrpn_patch_merge(const void *src, void *dst, int n, int gx, int gy, int gz, int c, int backward, int dtype,
                     nrpn_stream_t stream);
/* shifted_window_attention core (feature_extractor.py:424-530) for window 4x4x4, head_dim 32 (C == 32*heads):
 * qkv [N,X,Y,Z,3C] (output of the qkv Linear on the un-padded tokens), qkv_bias f32 [3C] (the value padded tokens take,
 * may be NULL), bias_table f32 [343,heads], rel_index i32 [64*64]; shift != 0 selects the shifted-window variant
 * (shift 2 on every axis longer than one window, -100 mask between regions).  out [N,X,Y,Z,C] feeds the proj Linear.
 * Backward overwrites dqkv [N,X,Y,Z,3C], dtable f32 [343,heads] and dbias_pad f32 [3C] (gradient reaching the qkv bias
 * through padded tokens; may be NULL).  Every (window, head) unit writes its partial bias-table gradient to `workspace`
 * (nrpn_window_attn_bwd_workspace_bytes) and a second kernel sums them: no same-address atomics, deterministic. */
/* bf16 tensors run on MFMA kernels that compute the relative-position index as code(i) - code(j) + 171 (the reference's
 * define_relative_position_index for a 4x4x4 window) instead of reading rel_index (fp32 always runs the VALU kernels; tools can force them
 * for bf16 through nerfrpn_tools.h). */
int nrpn_window_attn_fwd(const void *qkv, const float *qkv_bias, const float *bias_table, const int32_t *rel_index, void *out,
                         int n, int gx, int gy, int gz, int c, int heads, int shift, int dtype, nrpn_stream_t stream);
size_t nrpn_window_attn_bwd_workspace_bytes(int n, int gx, int gy, int gz, int heads);
int nrpn_window_attn_bwd(const void *qkv, const float *qkv_bias, const float *bias_table, const int32_t *rel_index,
                         const void *dout, void *dqkv, float *dtable, float *dbias_pad, int n, int gx, int gy, int gz, int c,
                         int heads, int shift, int dtype, int accumulate_table, void *workspace, nrpn_stream_t stream);   /* != 0: dtable += */

/* ------------------------------------------------------------------------------------------------
 * FCOS variant of the path.  [a23]  (model/fcos/fcos.py, inference.py, loss.py, utils.py)
 *   Flattened location order everywhere: level-major, then scene, then voxel (x, y, z; z fastest) -- the order of
 *   box_cls[l].permute(0,2,3,4,1).reshape(-1) concatenated over levels (loss.py:506-518).  Locations are index
 *   arithmetic: idx * stride + stride / 2 (fcos.py:233-250).  dims: host int32 [levels*3]; strides: host int32 [levels];
 *   ori_sizes: host f32 [n*3] un-padded scene sizes (NULL = no padding mask, as the reference does for batch 1).
 * ---------------------------------------------------------------------------------------------- */
/* nn.GroupNorm(groups, C) (+ fused ReLU) on channels-last [n, rows, C] (tower norm, fcos.py:57,69); mean/rstd f32 [n*groups]
 * are saved for the backward, which overwrites dgamma/dbeta.  workspace: nrpn_groupnorm_workspace_bytes(n, c, groups). */
size_t nrpn_groupnorm_workspace_bytes(int n, int c, int groups);
int nrpn_groupnorm_fwd(const void *x, void *y, const float *gamma, const float *beta, float *mean, float *rstd, int n,
                       int64_t rows, int c, int groups, float eps, int relu, int dtype, void *workspace, nrpn_stream_t stream);
int nrpn_groupnorm_bwd(const void *x, const void *y, const void *dy, void *dx, const float *gamma, const float *mean,
                       const float *rstd, float *dgamma, float *dbeta, int n, int64_t rows, int c, int groups, int relu, int dtype,
                       int accumulate_params, void *workspace, nrpn_stream_t stream);   /* accumulate_params != 0: dgamma / dbeta += */
/* Head epilogue for one level (fcos.py:104-128).  cls_out / box_out: f32 [rows, wrows] outputs of the fused 3x3x3 GEMMs
 * (cls_out col 0 = cls_logits, col 1 = centerness when !ctr_on_reg; box_out cols 0..reg_dim-1 = bbox_pred, col reg_dim =
 * centerness when ctr_on_reg).  reg = norm_reg ? [relu(scale*raw[:6]) * stride_mul, scale*raw[6:]] : exp(scale*raw).
 * The backward overwrites d_cls_out / d_box_out and writes d_scale[0]; d_scale points to nrpn_fcos_reduce_floats() floats whose
 * first two are ZERO on entry ([0] result, [1] ticket -- left zero --, then one partial per workgroup, summed in workgroup order: no
 * floating-point atomics, the result does not depend on scheduling). */
int nrpn_fcos_reduce_floats(void);
int nrpn_fcos_head_out_f32(const float *cls_out, const float *box_out, int wrows, const float *scale, float stride_mul,
                           int norm_reg, int reg_dim, int ctr_on_reg, int64_t rows, float *logits, float *reg, float *ctr,
                           nrpn_stream_t stream);
int nrpn_fcos_head_out_bwd_f32(const float *box_out, int wrows, const float *scale, float stride_mul, int norm_reg, int reg_dim,
                               int ctr_on_reg, int64_t rows, const float *d_logits, const float *d_reg, const float *d_ctr,
                               float *d_cls_out, float *d_box_out, float *d_scale, nrpn_stream_t stream);
/* Per-GT summary [count, 8] = footprint AABB (6) + midpoint offsets alpha, beta of encode_fcos_obb (utils.py:65-105);
 * width 6 (AABB GT) copies the box and zeroes alpha/beta. */
int nrpn_fcos_gt_summary_f32(const float *gt, int count, int width, float *summary, nrpn_stream_t stream);
/* FCOSLossComputation.prepare_targets (loss.py:270-437): per location the smallest-volume GT among those passing centre
 * sampling (radius * stride; radius <= 0: inside the box) and the level's size range; labels i8 {1, 0, -1 = padding},
 * reg_targets f32 [total, reg_dim] (first 6 / stride when norm_reg), num_pos = number of label-1 locations.
 * gt_offsets: host int32 [n+1] rows of `summary` per scene. */
int nrpn_fcos_targets_f32(const float *summary, const int32_t *gt_offsets, int n, int levels, const int32_t *dims,
                          const int32_t *strides, const float *ori_sizes, float radius, int norm_reg, int reg_dim, int8_t *labels,
                          float *reg_targets, int32_t *num_pos, nrpn_stream_t stream);
/* torchvision sigmoid_focal_loss(alpha, gamma=2, reduction='sum') over labels >= 0 (loss.py:541-545): loss_sum[0] (overwritten;
 * loss_sum points to nrpn_fcos_reduce_floats() floats of scratch, ordered workgroup partials as above) and, when dlogits != NULL,
 * d loss_sum / d logit per element. */
int nrpn_fcos_focal_f32(const float *logits, const int8_t *labels, int64_t count, float alpha, float *loss_sum, float *dlogits,
                        nrpn_stream_t stream);
/* FCOSPostProcessor.forward_for_single_feature_map, all levels at once (inference.py:56-88): score = sigmoid(cls) *
 * sigmoid(ctr) where the location is un-padded and sigmoid(cls) > pre_nms_thresh, else -1. */
int nrpn_fcos_scores_f32(const float *logits, const float *ctr, int n, int levels, const int32_t *dims, const int32_t *strides,
                         const float *ori_sizes, float pre_nms_thresh, float *scores, nrpn_stream_t stream);
/* Decode the top-k candidates of every (level, scene) segment (inference.py:108-133): idx i32 [levels*n*seg_len] location
 * index inside the segment (< 0: empty slot), score = cls*ctr.  AABB: loc -/+ reg clipped to the scene (ori_sizes required);
 * OBB: decode_fcos_obb (utils.py:12-62).  out_scores = sqrt(score), or -1 for empty slots / boxes below min_size. */
int nrpn_fcos_decode_f32(const int32_t *idx, const float *score, int64_t count, int64_t seg_len, const float *reg, int n,
                         int levels, const int32_t *dims, const int32_t *strides, const float *ori_sizes, int reg_dim,
                         float min_size, float *boxes, float *out_scores, float *out_levels, nrpn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Rotated 3D RoIAlign of the second-stage detector.  [f3]  Replaces the reference's second native op, pybind module rotated_roi_3d:
 *   roi_align_rotated_3d_forward(input f32[N,C,W,L,H], rois f32[R,8], spatial_scale, pw, pl, ph, sampling_ratio) -> [R,C,pw,pl,ph]
 *   roi_align_rotated_3d_backward(grad, rois, spatial_scale, pw, pl, ph, N, C, W, L, H, sampling_ratio) -> [N,C,W,L,H]
 *   (model/rotated_align/src/vision_3d.cpp, cuda_3d/ROIAlignRotated3D_cuda.cu:78-343, roi_align_rotate_3d.py:13-58).
 *   rois rows = (batch index, cx, cy, cz, w, l, h, theta in DEGREES); sampling_ratio <= 0: ceil(roi extent / pooled extent) samples.
 * Here: channels-last feature map feat [N][X][Y][Z][C] (dtype), out / grad_out [R][pw][pl][ph][C] (dtype), C % 4 == 0.
 * The backward accumulates through 64-bit fixed-point integer atomics in `workspace` (nrpn_roi_align_rotated_3d_bwd_workspace_bytes,
 * zeroed by the call) and converts once: deterministic, unlike the reference's fp32 atomicAdd.
 * ---------------------------------------------------------------------------------------------- */
int nrpn_roi_align_rotated_3d_fwd(const void *feat, const float *rois, int num_rois, int n, int x, int y, int z, int c,
                                  float spatial_scale, int pw, int pl, int ph, int sampling_ratio, void *out, int dtype,
                                  nrpn_stream_t stream);
size_t nrpn_roi_align_rotated_3d_bwd_workspace_bytes(int n, int x, int y, int z, int c);
int nrpn_roi_align_rotated_3d_bwd(const void *grad_out, const float *rois, int num_rois, int n, int x, int y, int z, int c,
                                  float spatial_scale, int pw, int pl, int ph, int sampling_ratio, void *grad_in, void *workspace,
                                  int dtype, nrpn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * ROIPool without the op -- the reference CLI's DEFAULT second-stage pooling (detector.py ROIPool(use_cuda=False)).  [f3]
 *   One launch per pyramid level over ALL RoIs of a scene: level_of [R] names each RoI's level, RoIs of other levels are skipped by the
 *   kernel (no host-side grouping).  feat: channels-last map [X][Y][Z][C] of that level, f32 | bf16; out: f32 [R][o0][o1][o2][C] (rows of
 *   other levels untouched); argmax: i32, same shape (max-pool forms).  Backward: dfeat [X][Y][Z][C] in feat's dtype = the gradient of THIS
 *   level from all its RoIs (deterministic: 64-bit fixed-point atomics), workspace nrpn_roipool_bwd_workspace_bytes.
 *   aabb (normal_forward, detector.py:397-438): crop [R][6] = (start x, y, z, size x, y, z) in level voxels -- the python slice
 *     [floor(lo / s), floor(hi / s)] after clipping; adaptive max-pool with zero padding at the high side (:377-384): kernel = stride =
 *     ceil(size / output); ties: first element in scan order (torch max_pool3d).  size <= 0 pools to zeros (the reference raises).
 *   obb (:264-395): rois [R][7] = (x, y, z, w, l, h, theta) in input voxels with the extents already enlarged; scale = input voxels per level
 *     voxel; grid of ceil(extent / scale) points rotated about the centre, each the reference's 8-corner blend
 *     sum feat[corner] * (1 - |dx||dy||dz|) / 8 (zero outside the map), then the adaptive max-pool (interpolation = 0) or a trilinear resize
 *     with aligned corners (interpolation = 1, :385-393).
 * ---------------------------------------------------------------------------------------------- */
size_t nrpn_roipool_bwd_workspace_bytes(int x, int y, int z, int c);
int nrpn_roipool_aabb_fwd(const void *feat, int x, int y, int z, int c, const int32_t *crop, const int32_t *level_of, int level, int num_rois,
                          int o0, int o1, int o2, float *out, int32_t *argmax, int dtype, nrpn_stream_t stream);
int nrpn_roipool_aabb_bwd(const float *dout, const int32_t *argmax, const int32_t *crop, const int32_t *level_of, int level, int num_rois,
                          int x, int y, int z, int c, int o0, int o1, int o2, void *dfeat, void *workspace, int dtype, nrpn_stream_t stream);
int nrpn_roipool_obb_fwd(const void *feat, int x, int y, int z, int c, const float *rois, const int32_t *level_of, int level, int num_rois,
                         float scale, int interpolation, int o0, int o1, int o2, float *out, int32_t *argmax, int dtype, nrpn_stream_t stream);
int nrpn_roipool_obb_bwd(const float *dout, const int32_t *argmax, const float *rois, const int32_t *level_of, int level, int num_rois,
                         float scale, int interpolation, int x, int y, int z, int c, int o0, int o1, int o2, void *dfeat, void *workspace,
                         int dtype, nrpn_stream_t stream);

/* Reduction step of the trainer's bf16 all-to-all gradient exchange (engine.FlatTrainer, exchange "a2a_bf16"): recv = bf16 [world][chunk]
 * (chunk `rank` of every peer's bucket), local = this rank's own fp32 chunk; out[i] = bf16(sum over ranks in ascending order, fp32
 * accumulation, the local chunk taken from `local`).  chunk % 4 == 0. */
int nrpn_a2a_reduce_bf16(const void *recv, const float *local, int rank, int world, int64_t chunk, void *out, nrpn_stream_t stream);
/* ------------------------------------------------------------------------------------------------
 * Optimiser step on a flat fp32 arena.  [a21]  (clip_grad_norm_ + AdamW, run_rpn.py:345-349,390-395)
 *   grad_scale folds the 1/world_size of the data-parallel mean into both kernels (sum all-reduce, no extra pass).
 *   sumsq: f32 device buffer of nrpn_grad_sumsq_floats() elements; [0] = sum((g*grad_scale)^2), the rest is scratch of the
 *   deterministic two-level reduction (fixed-size block partials summed in index order: bit-identical from run to run);
 *   step applies g *= grad_scale * min(1, max_norm/(sqrt(sumsq[0])+1e-6)),
 *   then decoupled-weight-decay Adam with bias correction (torch.optim.AdamW semantics).
 * ---------------------------------------------------------------------------------------------- */
int nrpn_grad_sumsq_floats(void);
int nrpn_grad_sumsq(const float *grad, int64_t count, float grad_scale, float *sumsq, nrpn_stream_t stream);
/* shadow_bf16 (optional, bf16 [count]): the updated parameters are also written as bf16 in the same element order -- for master
 * weights kept in the forward GEMM layout this IS the packed bf16 operand of the next forward (no repack pass). */
int nrpn_adamw_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t count,
                    const float *sumsq, float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int step, void *shadow_bf16, nrpn_stream_t stream);
/* the same, and the consumed gradient is cleared in the same pass (grad[i] = 0): the next backward accumulates into a clean arena */
int nrpn_adamw_step_zero_grad(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t count,
                              const float *sumsq, float grad_scale, float max_norm, float lr, float beta1, float beta2, float eps,
                              float weight_decay, int step, void *shadow_bf16, nrpn_stream_t stream);
/* GEMM-layout master weights inside a flat arena: forward layout [taps][Cout][Cin] per weight.
 * nrpn_transpose_weights: ONE launch writes the dgrad operand [taps reversed][Cin][Cout] (dtype) of every weight listed in the
 *   device-side job table (int64 [nweights][4] = arena element offset, taps, Cout, Cin; tile_prefix int32 [nweights+1] = running
 *   count of 64x64 tiles, total_tiles = tile_prefix[nweights]) at the same arena offsets of `dst`.
 * nrpn_reduce_slices: dst[i] (+)= sum over the S voxel-slice partials of a wgrad (already in the forward layout), in slice order. */
int nrpn_transpose_weights(const float *master, void *dst, const int64_t *table, const int32_t *tile_prefix, int nweights,
                           int total_tiles, int dtype, nrpn_stream_t stream);
/* bias_partials (optional): the per-slice column sums a nrpn_conv3d_wgrad call with NRPN_WGRAD_DEFER_BIAS left in its workspace at
 * byte offset nrpn_conv3d_wgrad_bias_offset(...) ([slices][wrows] f32); the same launch then also writes / accumulates gbias[cout]. */
int nrpn_reduce_slices(const float *partials, int slices, int64_t count, float *dst, int accumulate, const float *bias_partials,
                       int wrows, int cout, float *gbias, int accumulate_bias, nrpn_stream_t stream);
size_t nrpn_conv3d_wgrad_bias_offset(int n, int gx, int gy, int gz, int ksize);

#ifdef __cplusplus
}
#endif
#endif /* NERFRPN_H */
