/*
 * include/msda_hip.h -- C ABI of libmsda_hip.so (gfx950 / MI355X).
 *
 * Multi-scale deformable attention: sampling-point gather + bilinear
 * interpolation + attention-weight reduction, forward and backward.  This is the
 * drop-in boundary for the only native component of MeMOTR: the entry points
 * below are what the reference's extension module `MultiScaleDeformableAttention`
 * binds.  Paths cited are under the reference repository root.
 *
 *   msda_forward_*   replaces  ms_deform_attn_forward   (models/ops/src/ms_deform_attn.h:20-39,
 *                    models/ops/src/cuda/ms_deform_attn_cuda.cu:20-80, exported by
 *                    models/ops/src/vision.cpp:14)
 *   msda_backward_*  replaces  ms_deform_attn_backward  (models/ops/src/ms_deform_attn.h:41-61,
 *                    models/ops/src/cuda/ms_deform_attn_cuda.cu:83-153, vision.cpp:15)
 *
 * Conventions
 *   - Plain pointers and sizes only; no torch / ATen types.  All data pointers are
 *     DEVICE pointers to contiguous row-major arrays:
 *         value        (N, S, M, D)         S = sum_l H_l*W_l
 *         shapes_dev   (L, 2) int64         (H_l, W_l)             [.cu:67,138]
 *         lstart_dev   (L,)   int64         first row of level l    [.cu:68,139]
 *         loc          (N, Lq, M, L, P, 2)  (x, y) normalised to [0,1] incl. padding
 *         attn         (N, Lq, M, L, P)
 *         out          (N, Lq, M*D)
 *         grad_out     (N, Lq, M*D)
 *         grad_value / grad_loc / grad_attn  shaped like value / loc / attn
 *   - `shapes_host` (L,2 int64, HOST memory) is optional (may be NULL).  When given
 *     it must equal shapes_dev; it lets the library plan level-aware launches
 *     without a device->host read.  Results never depend on it.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls are
 *     asynchronous, never synchronise and are thread-safe.  Global state: the tuning
 *     options below and, for self-attention over the pyramid, one 8-KiB device +
 *     128-byte mapped-host record per (device, direction, call site, M, L, P,
 *     element size) -- NOT per geometry -- through which the kernels report how many
 *     sampling points left their windows (see "kernel selection"; all records of a
 *     device live in one block allocated on the first such call, never while
 *     `stream` is capturing).
 *   - Buffers are borrowed for the duration of the enqueued work.  `out`,
 *     `grad_loc`, `grad_attn` are fully overwritten.  `grad_value` is accumulated
 *     into with hardware float atomics: it must be zero on entry (the reference
 *     allocates it with at::zeros_like, .cu:121) unless `zero_grad_value` != 0, in
 *     which case the library enqueues the memset itself.
 *   - Return value: MSDA_OK (0) on success; a negative MSDA_E* code for argument
 *     errors; a positive hipError_t if the launch failed (the reference only
 *     printf()s those, ms_deform_im2col_cuda.cuh:948-952 -- here they are
 *     reported).  msda_last_error() returns a thread-local description.
 *   - Floating point: f32 and f64 as in the reference (AT_DISPATCH_FLOATING_TYPES,
 *     .cu:64,134).  The *_bf16 entry points are an extension with no reference
 *     counterpart: value / out / grad_out / grad_value-in-fp32 policy documented in
 *     DESIGN.md.
 *   - Integer/index arithmetic (floor of loc*size-0.5, corner indices, the
 *     (-1,H)x(-1,W) gate, zero padding per corner) is bit-identical to the
 *     reference kernels: the product loc*size rounded to float, then 0.5
 *     subtracted (.cuh:33-84,285-288; oracle built with -ffp-contract=off).
 *     Whether the reference BINARY contracts `.cuh:285` (`loc * size - 0.5`) into
 *     one fused multiply-add is UNDECIDED without nvcc: the literal 0.5 is a
 *     double, so for scalar_t = float the source reads fpext(mul) - 0.5 -> fptrunc,
 *     which a compiler may or may not narrow back to a float fma.  The library and
 *     the oracle follow the uncontracted source; tests/test_oracle_golden.py keeps
 *     the fused reading as a probe: zero of 2.86 M corner indices flip on either
 *     benchmark distribution (17,063 of them do, by one pixel with a bilinear
 *     weight <= 2^-23, for locations placed exactly on pixel centres).
 *     msda_sample_indices_f32 exposes the arithmetic for parity tests.
 */
#ifndef MSDA_HIP_H_
#define MSDA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSDA_OK 0
#define MSDA_EINVAL (-1)   /* null pointer / non-positive dimension               */
#define MSDA_ERANGE (-2)   /* a tensor is too large for 32-bit index arithmetic   */
#define MSDA_ENOTSUP (-3)  /* dtype/shape combination not supported (bf16: D%8)   */

/* ABI version: bumped on any signature change. */
int msda_abi_version(void);

/* Thread-local, never NULL; empty string when the last call succeeded. */
const char *msda_last_error(void);

/* ---- forward: replaces ms_deform_attn_forward (ms_deform_attn.h:20-39) ---- */
int msda_forward_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                     const float *loc, const float *attn,
                     int N, int S, int M, int D, int L, int Lq, int P,
                     float *out, const int64_t *shapes_host, void *stream);

int msda_forward_f64(const double *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                     const double *loc, const double *attn,
                     int N, int S, int M, int D, int L, int Lq, int P,
                     double *out, const int64_t *shapes_host, void *stream);

/* value/out are bf16 (uint16 storage), loc/attn fp32, accumulation fp32. */
int msda_forward_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                      const float *loc, const float *attn,
                      int N, int S, int M, int D, int L, int Lq, int P,
                      uint16_t *out, const int64_t *shapes_host, void *stream);

/* ---- backward: replaces ms_deform_attn_backward (ms_deform_attn.h:41-61) ---- */
int msda_backward_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                      const float *loc, const float *attn, const float *grad_out,
                      int N, int S, int M, int D, int L, int Lq, int P,
                      float *grad_value, float *grad_loc, float *grad_attn,
                      int zero_grad_value, const int64_t *shapes_host, void *stream);

int msda_backward_f64(const double *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                      const double *loc, const double *attn, const double *grad_out,
                      int N, int S, int M, int D, int L, int Lq, int P,
                      double *grad_value, double *grad_loc, double *grad_attn,
                      int zero_grad_value, const int64_t *shapes_host, void *stream);

/* value/grad_out bf16; grad_value accumulated in fp32 (N,S,M,D floats); grad_loc/grad_attn fp32. */
int msda_backward_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                       const float *loc, const float *attn, const uint16_t *grad_out,
                       int N, int S, int M, int D, int L, int Lq, int P,
                       float *grad_value, float *grad_loc, float *grad_attn,
                       int zero_grad_value, const int64_t *shapes_host, void *stream);

/* The same backward with caller-provided scratch (ABI 6; device memory, contents undefined on entry and exit).
 * msda_backward_workspace_bytes(fused, ...) is what the NEXT backward call of this thread's call site can use: the
 * fused prologue's block (fused != 0: msda_fused_workspace_bytes) plus -- when that call would build grad_value by
 * sort + gather instead of float atomics ("bwd_variant" 13, or kernel-selection level 2: self-attention over the
 * pyramid whose sampling points land far from their queries; memotr_amd/csrc/msda_bwd_sorted.h) -- 8 bytes per
 * (query, head, point, corner) of records and a few tables; 0 when the call needs none.  `elem_bytes` is the size of a
 * value element (4 / 2), `stream` the stream the call will go to (a capturing stream takes the level the last poll
 * announced).  workspace == NULL or too small: exactly msda_backward_* (whole-row float atomics). */
size_t msda_backward_workspace_bytes(int fused, int N, int S, int M, int D, int L, int Lq, int P, int elem_bytes,
                                     void *stream);

int msda_backward_ws_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                         const float *loc, const float *attn, const float *grad_out,
                         int N, int S, int M, int D, int L, int Lq, int P,
                         float *grad_value, float *grad_loc, float *grad_attn,
                         int zero_grad_value, const int64_t *shapes_host,
                         void *workspace, size_t workspace_bytes, void *stream);

int msda_backward_ws_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                          const float *loc, const float *attn, const uint16_t *grad_out,
                          int N, int S, int M, int D, int L, int Lq, int P,
                          float *grad_value, float *grad_loc, float *grad_attn,
                          int zero_grad_value, const int64_t *shapes_host,
                          void *workspace, size_t workspace_bytes, void *stream);

/* ---- fused prologue: replaces the elementwise chain of the reference MODULE between its query projections
 * and the operator (models/ops/modules/ms_deform_attn.py:104-123) ----
 * Instead of materialised sampling locations and attention weights the kernels take
 *     proj      (N*Lq, proj_stride) fp32   one row per query: [offsets (M,L,P,2) | logits (M,L,P)], i.e. the raw
 *                                          outputs of `sampling_offsets` and `attention_weights` side by side
 *                                          (proj_stride >= 3*M*L*P elements)
 *     ref       (N*Lq, L, ref_dim) fp32    reference points, ref_dim = 2 (x, y) or 4 (x, y, w, h)
 *     pad_mask  (N, S) uint8 or NULL       non-zero = padded pixel: its `value` row reads as zero and receives no
 *                                          gradient (= value.masked_fill(mask, 0) of ms_deform_attn.py:107-108)
 * and compute in-kernel, per (query, head): attention weights = softmax over the L*P logits (:111), locations =
 * ref + off / (W_l, H_l) for ref_dim 2 (:114-117) or ref_xy + off / P * ref_wh * 0.5 for ref_dim 4 (:118-120), each
 * step an IEEE fp32 operation in the reference's order -- the index arithmetic downstream sees the bits torch would
 * have produced (msda_fused_points_f32 exposes them).  L*P <= 64.
 * Backward: grad_value as msda_backward_*; grad_proj (N*Lq, proj_stride) receives d/d offsets and d/d logits (softmax
 * and location Jacobians applied; the 3*M*L*P used columns of every row are overwritten); grad_ref_part, when not
 * NULL, receives (N, Lq, M, L, ref_dim) per-head partial sums of d/d reference points (the caller sums over M).
 * Returns MSDA_ENOTSUP for L*P > 64. */
int msda_fused_forward_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                           const float *proj, int proj_stride, const float *ref, int ref_dim,
                           const uint8_t *pad_mask,
                           int N, int S, int M, int D, int L, int Lq, int P,
                           float *out, const int64_t *shapes_host, void *stream);

int msda_fused_forward_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                            const float *proj, int proj_stride, const float *ref, int ref_dim,
                            const uint8_t *pad_mask,
                            int N, int S, int M, int D, int L, int Lq, int P,
                            uint16_t *out, const int64_t *shapes_host, void *stream);

int msda_fused_backward_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                            const float *proj, int proj_stride, const float *ref, int ref_dim,
                            const uint8_t *pad_mask, const float *grad_out,
                            int N, int S, int M, int D, int L, int Lq, int P,
                            float *grad_value, float *grad_proj, float *grad_ref_part,
                            int zero_grad_value, const int64_t *shapes_host, void *stream);

int msda_fused_backward_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                             const float *proj, int proj_stride, const float *ref, int ref_dim,
                             const uint8_t *pad_mask, const uint16_t *grad_out,
                             int N, int S, int M, int D, int L, int Lq, int P,
                             float *grad_value, float *grad_proj, float *grad_ref_part,
                             int zero_grad_value, const int64_t *shapes_host, void *stream);

/* The same backward with a caller-provided scratch buffer of msda_fused_workspace_bytes(N, Lq, M, L, P) bytes
 * (device memory, contents undefined on entry and exit).  With it the region-tiled path runs as three kernels --
 * the prologue once per row into the workspace, the plain tiled kernel, the Jacobians in place in grad_proj --
 * instead of redoing the row softmax and the location arithmetic in each of the L workgroups a region takes
 * (298 -> ~250 us at the encoder shape).  workspace == NULL or too small: exactly msda_fused_backward_*.
 * A workspace of msda_backward_workspace_bytes(1, ...) bytes (>= msda_fused_workspace_bytes) also admits the
 * sort + gather form of grad_value when the call site's sampling points land far from their queries. */
size_t msda_fused_workspace_bytes(int N, int Lq, int M, int L, int P);

int msda_fused_backward_ws_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                               const float *proj, int proj_stride, const float *ref, int ref_dim,
                               const uint8_t *pad_mask, const float *grad_out,
                               int N, int S, int M, int D, int L, int Lq, int P,
                               float *grad_value, float *grad_proj, float *grad_ref_part,
                               int zero_grad_value, const int64_t *shapes_host,
                               void *workspace, size_t workspace_bytes, void *stream);

int msda_fused_backward_ws_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                                const float *proj, int proj_stride, const float *ref, int ref_dim,
                                const uint8_t *pad_mask, const uint16_t *grad_out,
                                int N, int S, int M, int D, int L, int Lq, int P,
                                float *grad_value, float *grad_proj, float *grad_ref_part,
                                int zero_grad_value, const int64_t *shapes_host,
                                void *workspace, size_t workspace_bytes, void *stream);

/* The fused backward given the forward's OUTPUT of the same call (ABI 6; `fwd_out` (N, Lq, M*D), value's dtype; NULL:
 * exactly msda_fused_backward_ws_*).  The softmax Jacobian's sum over a row's points, sum_j a_j dL/da_j, equals
 * <grad_out_row, out_row> (out = sum_j a_j sampled_j), so a kernel that owns one pyramid level of a region can finish its
 * logits' gradients without the other levels' results: for the encoder's calls (fp32, L*P = 16, 2-d reference points,
 * 16-byte aligned projection rows) the default backward is then ONE kernel -- no attention-weight kernel in front, no
 * Jacobian kernel behind, no workspace (150 -> ~135 us at the encoder shape).  An autograd function passes the tensor its
 * forward returned (the next layer's Linear keeps it alive anyway).  Other calls ignore it. */
int msda_fused_backward_out_f32(const float *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                                const float *proj, int proj_stride, const float *ref, int ref_dim,
                                const uint8_t *pad_mask, const float *grad_out, const float *fwd_out,
                                int N, int S, int M, int D, int L, int Lq, int P,
                                float *grad_value, float *grad_proj, float *grad_ref_part,
                                int zero_grad_value, const int64_t *shapes_host,
                                void *workspace, size_t workspace_bytes, void *stream);

int msda_fused_backward_out_bf16(const uint16_t *value, const int64_t *shapes_dev, const int64_t *lstart_dev,
                                 const float *proj, int proj_stride, const float *ref, int ref_dim,
                                 const uint8_t *pad_mask, const uint16_t *grad_out, const uint16_t *fwd_out,
                                 int N, int S, int M, int D, int L, int Lq, int P,
                                 float *grad_value, float *grad_proj, float *grad_ref_part,
                                 int zero_grad_value, const int64_t *shapes_host,
                                 void *workspace, size_t workspace_bytes, void *stream);

/* ---- `value` as a slice of a wider tensor (round 6, ABI 7) ----
 * The reference operator takes a contiguous (N, S, M, D) `value` (ms_deform_attn_cuda.cu:30, AT_ASSERTM contiguous): six
 * decoder layers that attend to the same memory each project it on their own (reference
 * models/deformable_decoder.py:303-310 -> ms_deform_attn.py:104: six (S x 256) x (256 x 256) GEMMs per frame, forward;
 * twelve and five full-size gradient additions backward).  One GEMM with the six weights stacked does the same work in
 * one launch each way -- if every layer can read ITS 256 columns of the (S x 1536) product in place:
 *   msda_next_value_pixel_stride(e)  the NEXT forward / backward call of the calling thread (any entry point above)
 *                        addresses pixel s of batch b at value + (b * S + s) * e (+ m * D + c) instead of
 *                        (b * S + s) * M * D; in a backward call `grad_value` has the same layout (the slice of the wide
 *                        gradient tensor that belongs to this layer) and `zero_grad_value` must be 0: the caller zeroes
 *                        the wide tensor once.  e >= M * D, e * sizeof(element) a multiple of 16; 0 or M * D: contiguous.
 *                        D = 32 float32 / bfloat16 calls only (the direct-gather forward, the row backward: the kernels the
 *                        decoder's calls take anyway); anything else returns MSDA_ENOTSUP and the caller passes a
 *                        contiguous copy.  The setting is consumed by that one call, successful or not. */
int msda_next_value_pixel_stride(long elements);

/* ---- kernel selection (round 4, records reworked in round 5 = ABI 5, per-site poll = ABI 6; memotr_amd/csrc/msda_select.h) ----
 * The cost of the reference kernels does not depend on where the sampling points land
 * (ms_deform_im2col_cuda.cuh:237-403); the windowed kernels here are fast for points near their query and slow
 * for points far away.  With "fwd_variant" / "bwd_variant" 0 the library therefore follows the data: the windowed
 * kernels report the share of corners outside their windows (cumulative counters: launches replayed from a hipGraph
 * count like eager ones), and the next call of the same call site picks the window margin -- or a kernel without
 * windows -- accordingly.  A record belongs to (device, direction, call site, M, L, P, element size), not to a
 * geometry; all records share one allocation per device.  Results never depend on the choice.
 *   msda_set_call_site   tags the calling thread's following calls (e.g. one tag per attention module; 0 = untagged),
 *                        so that modules with different learnt offsets on the same geometry are judged separately;
 *   msda_selector_last   level (0 = small windows ... top = no windows) and the last measured shares of corners outside
 *                        the window / outside the next smaller window, for this thread's last selected call
 *                        (shares < 0: nothing measured yet);
 *   msda_selector_next   the transition function itself (pure; for tests): next level from (kind 0 fwd / 1 bwd,
 *                        level, off-window and inner-window shares in 1/1000);
 *   msda_selector_poll   for callers that replay captured launches (a hipGraph makes no library call per launch): reads
 *                        every record of the current device, moves the levels, and stores a signature of the levels a
 *                        call would run at now (0: every record at level 0).  A graph cache keyed on the signature
 *                        replays the graph captured at those levels and captures another when a level has moved; a call
 *                        made while its stream is capturing takes the level the last poll announced.  Every 32nd poll
 *                        announces one level down for records at a level without windows (the probe that lets a record
 *                        come back).  Returns the number of records read;
 *   msda_selector_poll_sites   (ABI 6) the same for the records of the `n_sites` call sites in `sites` only: a graph
 *                        cache hashes the levels of the modules ITS graphs hold, so a move or a probe of another
 *                        module's record does not change its key, and the signature does not depend on which slot of
 *                        the table a site occupies.  `probe` != 0: records at a level without windows announce one level
 *                        down (the caller counts its own polls).  Once a record has been polled, eager calls no longer
 *                        overwrite what the poll announced (a cache's eager warm-up calls ahead of a capture used to);
 *   msda_selector_reset  forgets every record of every device (waits for the device, clears the blocks; captured graphs
 *                        stay valid -- the blocks are not freed).  For processes that retire a model and build another,
 *                        and for test suites: records of modules that no longer run still take part in the signature. */
void msda_set_call_site(uint64_t site);
int msda_selector_last(int *level, float *off_share, float *inner_share);
int msda_selector_next(int kind, int level, int off_permille, int inner_permille);
int msda_selector_poll(uint64_t *signature);
int msda_selector_poll_sites(const uint64_t *sites, int n_sites, int probe, uint64_t *signature);
int msda_selector_reset(void);

/* ---- parity hooks ----
 * msda_sample_indices_f32: the integer side of the sampling arithmetic.  For every (n,q,m,l,p):
 * h_low = floor(loc_y*H_l - 0.5), w_low = floor(loc_x*W_l - 0.5) and gate = (-1 < h < H_l && -1 < w < W_l) as computed
 * by the device function the kernels use (.cuh:285-288, :38-39).  Outputs are (N,Lq,M,L,P) int32/int32/uint8.
 * msda_fused_points_f32: the fused prologue on its own -- loc_out (N,Lq,M,L,P,2) and attn_out (N,Lq,M,L,P) exactly as
 * the fused kernels compute them from proj / ref. */
int msda_sample_indices_f32(const int64_t *shapes_dev, const float *loc,
                            int N, int M, int L, int Lq, int P,
                            int32_t *h_low, int32_t *w_low, uint8_t *gate, void *stream);

int msda_fused_points_f32(const int64_t *shapes_dev, const float *proj, int proj_stride, const float *ref,
                          int ref_dim, int N, int M, int L, int Lq, int P,
                          float *loc_out, float *attn_out, void *stream);

/* ---- tuning knobs (benchmarks / tests only; defaults pick the fastest correct path) ----
 * key: "fwd_variant" (0 = auto, 1 = generic one-thread-per-output kernel, 3 = D = 32 gather, 12 = windowed forward) |
 *       "bwd_variant" (0 = auto, 1 = generic, 10 = fixed-point windows, 12 = counting sort, 13 = global sort + gather
 *       through the caller's workspace), see DESIGN.md;
 *       "fwd_block" | "bwd_block" (threads per block, multiple of 64), "fwd_grid_mult" | "bwd_grid_mult" (blocks per
 *       CU), "bwd_tile_margin" (LDS window margin in pixels, variant 10), "bwd_split" (1 = the three-kernel fused
 *       backward when a workspace is given),
 *       "fwd_win_auto" (1 = fp32 self-attention over the pyramid takes the windowed forward, variant 12; 0 = the
 *       gather kernel), "fwd_win_rlog" / "fwd_win_rlogx" / "fwd_win_block" (log2 of the region height / width on the
 *       finest level and threads per workgroup; 0 = auto: 16 x 16 pixels, 512 threads), "fwd_win_margins" (per level as
 *       0xL3L2L1L0), "fwd_win_l0" (first windowed level), "fwd_win_bf16" (1: bf16 rows take the
 *       windowed forward too; measured slower than the gather kernel, default 0), "fwd_win_early" / "fwd_win_wps" (level-0 rows requested
 *       ahead, register budget; "fwd_win_wps" 2 with 256 threads: the 256-register build, twelve LDS points per wait),
 *       "fwd_win_grid" (1, default: the default shape becomes equal regions of any size when a per-XCD slot estimate
 *       says they fill the workgroup slots in fewer rounds than the power-of-two regions; 0: never),
 *       "fwd_win_rsy" / "fwd_win_rsx" (region height / width on the finest level in pixels: grid mode with exactly that
 *       size), "fwd_win_place" (1 = every workgroup measures its window placement, rounds 3-4;
 *       0 = from the call site's running means), profiling switches; tools/fwd_win_sweep.py lists them;
 *       "fwd_head_major" (head-major block numbering of the gather
 *       kernel), "bwd_rows" / "bwd_rows_block" (the 32-lanes-per-row backward of decoder-shaped calls), "bwd_soft" (1: the one-kernel fused backward when the forward's output is given; 0: the side kernels), "bwd_sorted" (1: selector
 *       level 2 builds grad_value by sort + gather when the caller gave scratch; 0: the rows kernel's float atomics),
 *       "bwd_wide_log2", "bwd_ablate" / "fwd_win_ablate" (profiling only: results are wrong by construction),
 *       "bwd_bins_margin" / "bwd_bins_margin_hi" / "bwd_bins_strip" (counting-sort backward, variant 12: window margins
 *       of selector levels 0 / 1, region rows per strip of the block walk), "auto_select" (0: no selection, level 0), "deterministic" (1: the
 *       FORWARD keeps one summation order whatever the selector's statistics say -- identical calls return identical bits,
 *       like the reference's atomic-free forward; the backward accumulates in an order that varies from run to run, like
 *       the reference's atomicAdd),
 *       "sel_level" (-1: follow the data; >= 0: pin the level), "sel_up0" / "sel_up1" / "sel_down1" / "sel_down2" /
 *       "sel_fwd_up" / "sel_fwd_down" (thresholds in 1/1000 of the valid corners; "sel_up1_rows" / "sel_down2_rows": the
 *       level 1 <-> 2 thresholds of call sites whose caller never asks msda_backward_workspace_bytes -- level 2 is then
 *       the rows kernel's float atomics, worth it only past ~10 %), "bwd_sort_qc" / "bwd_sort_emult" (sorted backward:
 *       queries per dots workgroup, chunks per emit workgroup; 0 = auto).
 * Returns MSDA_OK or MSDA_EINVAL for an unknown key / bad value. */
int msda_set_option(const char *key, int value);
int msda_get_option(const char *key, int *value);

/* Name of the kernel the last forward/backward call on this thread dispatched to
 * (for profiles and tests; never NULL). */
const char *msda_last_kernel(void);

#ifdef __cplusplus
}
#endif
#endif /* MSDA_HIP_H_ */
