/* clip_ops_hip.h -- C ABI of libclip_ops_hip.so: the small-tensor chains of the clip train step as single gfx950
 * kernels.
 *
 * The decoder / criterion side of a MeMOTR train step is hundreds of element-wise torch kernels on tensors of a
 * few thousand elements (matching cost, focal / L1 / GIoU losses, anchor embedding, box refinement).  On MI355X
 * each costs 5-7 us of queue time whatever its size, forward and again (2-3x) backward; these entry points
 * evaluate one whole chain per launch.  Arithmetic follows the reference formulas cited per function; all
 * tensors are fp32 (indices int64), device pointers, plain sizes and element strides -- no torch types.
 *
 * Every function returns 0 on success or a non-zero code (clipops_last_error() has the text) and launches on
 * `stream` (a hipStream_t passed as void*; NULL = the default stream).  Nothing here synchronises.
 */
#ifndef CLIP_OPS_HIP_H
#define CLIP_OPS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLIPOPS_ABI_VERSION 10

int clipops_abi_version(void);
const char *clipops_last_error(void);

/* Matching cost of `n_layers` decoder layers at once (reference models/matcher.py:83-121, focal-style class cost):
 *   cost[l,q,t] = w_bbox * |box[l,q] - gt_box[t]|_1 + w_class * (pos - neg)(sigmoid(logit[l,q,gt_label[t]]))
 *                 + w_giou * (-GIoU(xyxy(box[l,q]), xyxy(gt_box[t])))
 * logits: element (l,q,k) at logits[l*logit_sl + q*logit_sq + k]; boxes: (l,q,c) at boxes[l*box_sl + q*box_sq + c]
 * (cxcywh).  gt_labels (T) int64, gt_boxes (T,4) and cost (n_layers,Q,T) are contiguous. */
int clipops_match_cost_f32(const float *logits, long logit_sl, long logit_sq, const float *boxes, long box_sl,
                           long box_sq, const int64_t *gt_labels, const float *gt_boxes, int n_layers, int Q, int K,
                           int T, float w_class, float w_bbox, float w_giou, float *cost, void *stream);

/* L1 and GIoU loss of `n` (prediction, target) box pairs (reference models/criterion.py:417-440: l1_loss(reduction
 * none).sum(-1) and 1 - diag(generalized_box_iou(xyxy(pred), xyxy(tgt)))).
 * Prediction i is row  lay[i]*row_mul + row_add + qidx[i]  of `boxes` (rows of 4 floats, cxcywh); target i is row
 * gidx[i] of tgt_boxes (gidx == NULL: row i).  weight (n) may be NULL; both outputs are multiplied by it. */
int clipops_pair_box_loss_fwd_f32(const float *boxes, const int64_t *lay, const int64_t *qidx, long row_mul,
                                  long row_add, const float *tgt_boxes, const int64_t *gidx, const float *weight,
                                  int n, float *l1, float *giou_loss, void *stream);
/* Gradient of the above with respect to the prediction rows: grad_boxes (same row layout as `boxes`) receives
 * plain stores for the n rows (the pairs of one call address distinct rows); the caller zero-fills it. */
int clipops_pair_box_loss_bwd_f32(const float *boxes, const int64_t *lay, const int64_t *qidx, long row_mul,
                                  long row_add, const float *tgt_boxes, const int64_t *gidx, const float *weight,
                                  int n, const float *grad_l1, const float *grad_giou, float *grad_boxes,
                                  void *stream);

/* IoU of n box pairs, no gradient (reference models/criterion.py:354-367, the `iou` field of the tracks):
 * iou[i] = IoU(xyxy(boxes[i]), xyxy(tgt_boxes[gidx[i]]))  (gidx == NULL: row i); boxes (n,4) cxcywh contiguous. */
int clipops_pair_iou_f32(const float *boxes, const float *tgt_boxes, const int64_t *gidx, int n, float *iou,
                         void *stream);

/* ---- per-frame bookkeeping of the criterion (round 6, ABI 10) ----
 * Which ground truth a carried track owns, and which ground truths nobody owns (reference models/criterion.py:166-182:
 * the `gt_ids_to_idx` dict -- the LAST ground truth wins when a frame repeats an id -- and the `unmatched` list):
 *   matched_idx[i] = max { j : gt_ids[j] == track_ids[i] }  or -1;   gt_free[j] = 1.0 if no track carries gt_ids[j] else 0.0
 * (float: it travels to the host in front of the cost tensor).  Replaces seven torch launches (compare / arange / product
 * / amax / subtract / any / not) by one. */
int clipops_track_ownership_i64(const int64_t *track_ids, int n_tracks, const int64_t *gt_ids, int n_gt,
                                int64_t *matched_idx, float *gt_free, void *stream);
/* Classification targets of every decoder layer for the focal loss (reference models/criterion.py:300-330):
 *   labels (n_layers, n_det + n_tracks) = K (background);  columns n_det.. of the layers with late[l] != 0 = the label of the
 *   ground truth the track owns (matched_idx, -1: background);  then labels[lay[p], q[p]] = gt_labels[g[p]] for the n_pairs
 *   matched (layer, detect query, ground truth) triples.  Replaces eleven torch launches (fill / gather / index_put /
 *   compare / clamp / gather / two fills / two selects / slice copy) by one. */
int clipops_focal_labels_i64(const int64_t *lay, const int64_t *q, const int64_t *g, int n_pairs,
                             const int64_t *gt_labels, int n_gt, const int64_t *matched_idx, int n_tracks,
                             const uint8_t *late, int n_layers, int n_det, int K, int64_t *labels, void *stream);

/* Sigmoid focal loss of stacked layers (reference models/criterion.py:442-467, RetinaNet form): per layer l
 *   loss[l] = sum_q mean_k  a_t * ce * (1 - p_t)^gamma,   target one-hot of labels[l,q] (label == K: background).
 * logits element (l,q,k) at logits[l*sl + q*sq + k]; labels (n_layers,Nq) int64 contiguous; loss (n_layers).
 * One workgroup per layer with a fixed-order reduction: results are run-to-run identical. */
int clipops_focal_fwd_f32(const float *logits, long sl, long sq, const int64_t *labels, int n_layers, int Nq, int K,
                          float alpha, float gamma, float *loss, void *stream);
/* grad_logits (n_layers,Nq,K) contiguous = grad_loss[l] * d loss[l] / d logit. */
int clipops_focal_bwd_f32(const float *logits, long sl, long sq, const int64_t *labels, int n_layers, int Nq, int K,
                          float alpha, float gamma, const float *grad_loss, float *grad_logits, void *stream);

/* Sine embedding of box coordinates (reference models/utils.py:78-85, `pos_to_pos_embed`): pos (n,K) ->
 * out (n, K*F),  out[i, k*F + j] = (j even ? sin : cos)((pos[i,k] * scale) / dim_t[j]);  dim_t (F) is the
 * temperature ladder the caller computed once (passed in so that both sides use the same floats). */
int clipops_sine_embed_fwd_f32(const float *pos, const float *dim_t, long n, int K, int F, float scale, float *out,
                               void *stream);
/* grad_pos (n,K) = sum_j grad_out[i, k*F + j] * d out / d pos. */
int clipops_sine_embed_bwd_f32(const float *pos, const float *dim_t, long n, int K, int F, float scale,
                               const float *grad_out, float *grad_pos, void *stream);

/* logit with both odds clamped (reference utils/utils.py:61-74, `inverse_sigmoid`), n elements:
 *   y = log(clamp(x, eps, 1) / clamp(1 - x, eps, 1));  the backward applies torch's clamp masks. */
int clipops_inverse_sigmoid_fwd_f32(const float *x, long n, float eps, float *y, void *stream);
int clipops_inverse_sigmoid_bwd_f32(const float *x, const float *grad_y, long n, float eps, float *grad_x, void *stream);

/* Iterative box refinement of the decoder (reference models/deformable_decoder.py:139-149, 4-d references):
 *   out = sigmoid(delta + inverse_sigmoid(ref)), n elements.  Backward from the saved `out`; grad_ref may be NULL. */
int clipops_refine_boxes_fwd_f32(const float *delta, const float *ref, long n, float eps, float *out, void *stream);
int clipops_refine_boxes_bwd_f32(const float *out, const float *ref, const float *grad_out, long n, float eps,
                                 float *grad_delta, float *grad_ref, void *stream);

/* Column sums of a small row-major matrix: out[c] = sum_r x[r*cols + c] -- the bias gradient of a Linear over a few
 * hundred query rows.  torch's generic reduction takes 12-17 us for 310 x 256 .. 2048 on MI355X (one of ~400 such calls
 * per train step); this one tiles 32 columns x 8 row lanes per workgroup and sums in a fixed order. */
int clipops_colsum_f32(const float *x, long rows, int cols, float *out, void *stream);
/* First pass for tall matrices: partial[k*cols + c] = sum of rows [k*chunk_rows, (k+1)*chunk_rows) of column c,
 * k < ceil(rows / chunk_rows); clipops_colsum_f32 over `partial` finishes (fixed order end to end). */
int clipops_colsum_partial_f32(const float *x, long rows, int cols, int chunk_rows, float *partial, void *stream);
/* The same from bf16 storage (fp32 partial sums): bias gradients of the bf16 linears -- torch's own multi-block
 * reduction returned garbage inside replayed hipGraphs (round 3, profiles/r03_graph_memset_probe.txt). */
int clipops_colsum_partial_bf16(const uint16_t *x, long rows, int cols, int chunk_rows, float *partial, void *stream);

/* Multi-head self-attention over the decoder queries (reference models/deformable_decoder.py:245-249: the
 * nn.MultiheadAttention call with query = key = tgt + pos, value = tgt; here after the input projections):
 *   out[b,i,h,:] = sum_j softmax_j(scale * q[b,i,h,:] . k[b,j,h,:]  |  key_mask[b,j] == 0) * v[b,j,h,:]
 * head_dim is 32, L <= CLIPOPS_MHA_MAX_L; fp32, exact softmax (per-lane online softmax merged over the 8 lanes of a
 * query row), K and V of a head staged in LDS.  q / k / v are addressed as base + b*batch_stride + i*row_stride + h*32
 * (element strides), so the packed (B, L, 2E) projection of q and k needs no copy; key_mask (B, L) bytes, non-zero =
 * ignore that key, may be NULL; out (B, L, H*32) contiguous; lse (B, H, L) receives max + log(sum) per row for the
 * backward.  Replaces the AOTriton kernels behind F.scaled_dot_product_attention on this path. */
#define CLIPOPS_MHA_MAX_L 512
int clipops_mha_fwd_f32(const float *q, const float *k, const float *v, long q_bs, long q_rs, long k_bs, long k_rs,
                        long v_bs, long v_rs, const uint8_t *key_mask, int B, int H, int L, float scale, float *out,
                        float *lse, void *stream);
/* Gradients of the above.  grad_out (B, L, H*32) contiguous; grad_q / grad_k / grad_v are addressed like q / k / v
 * (each its own batch / row stride, so grad_q and grad_k can be the two halves of one (B, L, 2E) buffer).  Two kernels:
 * one by query rows (grad_q), one by key rows (grad_k, grad_v); probabilities are recomputed from lse. */
int clipops_mha_bwd_f32(const float *q, const float *k, const float *v, long q_bs, long q_rs, long k_bs, long k_rs,
                        long v_bs, long v_rs, const uint8_t *key_mask, const float *out, const float *lse,
                        const float *grad_out, int B, int H, int L, float scale, float *grad_q, long gq_bs, long gq_rs,
                        float *grad_k, long gk_bs, long gk_rs, float *grad_v, long gv_bs, long gv_rs, void *stream);

/* Residual add + LayerNorm over rows of 256 floats (the post-norm blocks of the encoder / decoder layers, reference
 * models/deformable_encoder.py:104-106, :126-127, models/deformable_decoder.py:251-252, :271-272, :311-312):
 *   sum = x + res;  y = (sum - mean) * rstd * gamma + beta   (mean / biased variance over the 256 columns, eps inside
 *   the sqrt).  One wavefront per row, 4 floats per lane.  `sum` (rows,256) and `stats` (rows,2) = (mean, rstd) are
 *   kept for the backward.  Replaces an element-wise add + native_layer_norm (2 kernels) forward. */
#define CLIPOPS_LN_COLS 256
int clipops_add_layer_norm_fwd_f32(const float *x, const float *res, const float *gamma, const float *beta, long rows,
                                   float eps, float *sum, float *y, float *stats, void *stream);
/* Backward: grad_sum (rows,256) = d/d(x) = d/d(res);  partial (ceil(rows/chunk_rows), 512) receives per-chunk column
 * sums of [grad_y * xhat | grad_y]; clipops_colsum_f32 over it yields [grad_gamma | grad_beta].  Replaces
 * layer_norm_grad_input + two gamma/beta kernels + the add's fan-out. */
int clipops_add_layer_norm_bwd_f32(const float *grad_y, const float *sum, const float *stats, const float *gamma,
                                   long rows, int chunk_rows, float *grad_sum, float *partial, void *stream);

/* Linear sum assignment of `n_problems` cost matrices of one shape (n_rows, n_cols) on the device -- what the
 * reference's matcher does on the host with scipy.optimize.linear_sum_assignment after a `.cpu()` copy
 * (models/matcher.py:122-124).  Same algorithm as scipy's (shortest augmenting paths, Crouse 2016, float64 arithmetic
 * on the float32 costs) with the same scan order and tie rule, so the PAIRS are scipy's, not just the cost
 * (memotr_amd/csrc/assign_core.h; held to scipy on 1000 random matrices with ties on the CPU and on the GPU).
 * Element (p, r, c) of `cost` is at cost[p*stride_problem + r*stride_row + c*stride_col] (element strides: a
 * (layers, Q, T) cost tensor and its transpose are both addressable without a copy).  One 64-lane wavefront per
 * problem, all state in LDS, max(n_rows, n_cols) <= CLIPOPS_ASSIGN_MAX_DIM.  Outputs, k = min(n_rows, n_cols):
 * row_ind, col_ind (n_problems, k) int32 -- pairs ordered by row like scipy's result; status (n_problems) int32,
 * may be NULL: k, or -1 for an infeasible matrix (every completion costs +inf: scipy raises ValueError).
 * NOT checked: NaN or -inf entries (scipy raises ValueError on them; this kernel returns some pairing) -- a caller
 * that cannot rule them out tests the matrix first.  Dynamic LDS is 13 n_r + 29 n_c bytes: past 64 KB (n ~ 1560) the
 * launcher raises the kernel's limit itself (round 4).
 * The solver and its parity tests; the model's criterion consumes scipy's result on the host (shelved). */
#define CLIPOPS_ASSIGN_MAX_DIM 2048
int clipops_assign_f32(const float *cost, long stride_problem, long stride_row, long stride_col, int n_problems,
                       int n_rows, int n_cols, int32_t *row_ind, int32_t *col_ind, int32_t *status, void *stream);

/* ---- the backbone's element-wise tail (round 4, ABI 9) ----
 * x = relu(x + shift[c] (+ res)) in place on a convolution's output, NCHW contiguous, `planes` = N * C planes of HW
 * elements; `res` (same shape) may be NULL.  Replaces, in ONE pass, what torchvision's Bottleneck.forward does with a
 * frozen batch norm folded into the convolution (reference models/backbone.py:70-76, FrozenBatchNorm2d :20-60): the
 * per-channel bias add, `out += identity` and the ReLU -- three passes over up to 344 MB per layer and clip.  The bf16
 * form rounds after the bias add and after the residual add like the kernels it replaces.  NaN propagates (x < 0 ? 0 : x). */
int clipops_shift_relu_f32(float *x, const float *shift, const float *res, long planes, int C, long HW, void *stream);
int clipops_shift_relu_bf16(uint16_t *x, const float *shift, const uint16_t *res, long planes, int C, long HW,
                            void *stream);

/* Backward of a query-sized Linear y = x W^T + b (x (rows, in), W (out, in), all contiguous fp32) in one launch:
 * grad_x (rows, in) = G' W, grad_w (out, in) = G'^T x, grad_b (out) = column sums of G', with G' = grad_y or, when the
 * forward fused a ReLU (`y_relu` = its output), grad_y masked by y > 0 -- torch.nn.functional.linear's backward as the
 * reference's decoder / heads / query updater run it a few hundred times per train step on 300-odd rows
 * (models/deformable_decoder.py, models/mlp.py, models/query_updater.py), where it is four dependent launches.  fp32 MFMA
 * (exact fp32 products and sums; the contraction order differs from a library GEMM's).  grad_x / grad_w / grad_b may be
 * NULL (grad_b needs grad_w). */
int clipops_linear_bwd_f32(const float *grad_y, const float *y_relu, const float *x, const float *w, int rows,
                           int in_features, int out_features, float *grad_x, float *grad_w, float *grad_b,
                           void *stream);

/* ... and its forward, y (rows, out) = [relu](x W^T + b) (`bias` may be NULL; in_features a multiple of 4): the same
 * 32 x 32 MFMA tiles with both operands read as 16-byte loads along the contraction. */
int clipops_linear_fwd_f32(const float *x, const float *w, const float *bias, int rows, int in_features,
                           int out_features, int relu, float *y, void *stream);

/* g2 = y > 0 ? g : 0 (the ReLU mask of a Linear whose forward fused the activation; torch's threshold_backward) AND
 * the per-chunk column sums of g2 (first pass of the bias gradient, finished by clipops_colsum_f32 over `partial`
 * (chunks, cols)) in one pass over (rows, cols) contiguous matrices: the encoder FFN's first linear
 * (models/deformable_encoder.py:100-103 of the reference) otherwise re-reads its 914 MB gradient for the sum: 522 vs
 * 637 us.  fp32 only: with 2-byte elements this tile shape loses to the two torch passes (546 vs 358 us, measured). */
int clipops_relu_bwd_colsum_partial_f32(const float *g, const float *y, long rows, int cols, int chunk_rows, float *g2,
                                        float *partial, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CLIP_OPS_HIP_H */
