"""Fused small-tensor chains of the clip train step: one gfx950 kernel per chain (include/clip_ops_hip.h) where the
reference issues a string of element-wise torch kernels.

Every function takes CUDA fp32 tensors and launches through the C ABI; ``*_reference`` are the element-wise torch
formulations of the same arithmetic (the reference's formulas) -- what CPU tensors run and what the GPU tests
compare the kernels with.  ``fused(t)`` decides: CUDA fp32 tensors use the kernels unless MEMOTR_FUSED_CLIP_OPS=0.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F


def fused(*tensors) -> bool:
    return (os.environ.get("MEMOTR_FUSED_CLIP_OPS", "1") != "0"
            and all(t.is_cuda and (t.dtype == torch.float32 or not t.is_floating_point()) for t in tensors))


def config_key() -> tuple:
    """The environment switches that change which kernels a captured graph contains."""
    return (os.environ.get("MEMOTR_FUSED_CLIP_OPS", "1"), os.environ.get("MEMOTR_COLSUM_TALL", "1"),
            os.environ.get("MEMOTR_ATTN_KERNELS", "1"), os.environ.get("MEMOTR_FUSED_LN", "1"),
            os.environ.get("MEMOTR_FUSED_SHIFT_RELU", "1"), os.environ.get("MEMOTR_FUSED_LINEAR_BWD", "1"),
            os.environ.get("MEMOTR_FUSED_LINEAR_FWD", "1"))


def _lib():
    from .. import _clip_lib        # raises ImportError when the library is missing: no silent substitute
    return _clip_lib


def _stream(t: torch.Tensor) -> int:
    """Launch stream of ``t``'s device.  The kernels launch on the CURRENT device (one device per process, as under
    DistributedDataParallel); tensors of another device would get that device's stream handle with the wrong device
    current, so that case is refused loudly instead of launching somewhere else."""
    idx = t.device.index
    if idx is not None and idx != torch.cuda.current_device():
        raise RuntimeError(f"clip_ops: tensor on cuda:{idx} but cuda:{torch.cuda.current_device()} is current; "
                           "call torch.cuda.set_device(tensor.device) (one device per process)")
    return torch.cuda.current_stream(t.device).cuda_stream


def _strides3(t: torch.Tensor):
    """(layer stride, row stride) in elements of a (n_layers, rows, C) view whose last axis is dense."""
    if t.dim() != 3 or (t.shape[2] > 1 and t.stride(2) != 1):
        raise RuntimeError("expected a (layers, rows, C) tensor with a dense last axis")
    return t.stride(0), t.stride(1)


# --------------------------------------------------------------------------------------------------------------
# matching cost (models/matcher.py of the reference, :83-121)
# --------------------------------------------------------------------------------------------------------------
def match_cost(logits: torch.Tensor, boxes: torch.Tensor, gt_labels: torch.Tensor, gt_boxes: torch.Tensor,
               w_class: float, w_bbox: float, w_giou: float, out: torch.Tensor = None) -> torch.Tensor:
    """(n_layers, Q, K) logits and (n_layers, Q, 4) cxcywh boxes (views allowed) against (T,) labels / (T, 4) boxes
    -> (n_layers, Q, T) cost.  ``out``: a contiguous float32 destination of n_layers * Q * T elements (any shape: the
    tail of the buffer that travels to the host)."""
    n_layers, Q, K = logits.shape
    T = gt_labels.shape[0]
    if out is not None:
        assert out.is_contiguous() and out.dtype == torch.float32 and out.numel() == n_layers * Q * T
        cost = out.view(n_layers, Q, T)
    else:
        cost = torch.empty((n_layers, Q, T), dtype=torch.float32, device=logits.device)
    lsl, lsq = _strides3(logits)
    bsl, bsq = _strides3(boxes)
    gt_labels, gt_boxes = gt_labels.contiguous(), gt_boxes.contiguous()
    L = _lib()
    L.check(L.lib.clipops_match_cost_f32(logits.data_ptr(), lsl, lsq, boxes.data_ptr(), bsl, bsq, gt_labels.data_ptr(),
                                         gt_boxes.data_ptr(), n_layers, Q, K, T, float(w_class), float(w_bbox),
                                         float(w_giou), cost.data_ptr(), _stream(logits)), "clipops_match_cost_f32")
    return cost


# --------------------------------------------------------------------------------------------------------------
# per-frame bookkeeping (include/clip_ops_hip.h, ABI 10): ownership of the ground truths, focal-loss targets
# --------------------------------------------------------------------------------------------------------------
def track_ownership(track_ids: torch.Tensor, gt_ids: torch.Tensor, free_out: torch.Tensor = None):
    """-> (matched_idx (n_tracks,) int64: index of the LAST ground truth carrying the track's id or -1,
    free (n_gt,) float32: 1 where no track carries the ground truth's id).  ``free_out``: where to write ``free``."""
    n_tr, n_gt = track_ids.shape[0], gt_ids.shape[0]
    matched = torch.empty((n_tr,), dtype=torch.int64, device=gt_ids.device)
    free = free_out if free_out is not None else torch.empty((n_gt,), dtype=torch.float32, device=gt_ids.device)
    assert free.is_contiguous() and free.dtype == torch.float32 and free.numel() == n_gt
    track_ids, gt_ids = track_ids.contiguous(), gt_ids.contiguous()
    L = _lib()
    L.check(L.lib.clipops_track_ownership_i64(track_ids.data_ptr(), n_tr, gt_ids.data_ptr(), n_gt, matched.data_ptr(),
                                              free.data_ptr(), _stream(gt_ids)), "clipops_track_ownership_i64")
    return matched, free


def track_ownership_reference(track_ids, gt_ids):
    n_gt = gt_ids.shape[0]
    eq = track_ids[:, None] == gt_ids[None, :]
    order = torch.arange(1, n_gt + 1, device=eq.device)
    return (eq * order).amax(1) - 1 if n_gt > 0 else torch.full_like(track_ids, -1), (~eq.any(0)).to(torch.float32)


def focal_labels(lay: torch.Tensor, q: torch.Tensor, g: torch.Tensor, gt_labels: torch.Tensor, matched_idx,
                 late: torch.Tensor, n_det: int, n_tracks: int, num_classes: int) -> torch.Tensor:
    """(n_layers, n_det + n_tracks) int64 focal-loss targets: background, the carried tracks' labels in the layers with
    ``late`` set, the matched (layer, query, ground truth) triples -- ``focal_labels_reference`` in one launch."""
    n_layers = late.shape[0]
    labels = torch.empty((n_layers, n_det + n_tracks), dtype=torch.int64, device=late.device)
    lay, q, g, gt_labels = lay.contiguous(), q.contiguous(), g.contiguous(), gt_labels.contiguous()
    L = _lib()
    L.check(L.lib.clipops_focal_labels_i64(lay.data_ptr(), q.data_ptr(), g.data_ptr(), lay.shape[0], gt_labels.data_ptr(),
                                           gt_labels.shape[0], None if matched_idx is None else matched_idx.contiguous().data_ptr(),
                                           n_tracks, late.contiguous().data_ptr(), n_layers, n_det, num_classes,
                                           labels.data_ptr(), _stream(late)), "clipops_focal_labels_i64")
    return labels


def focal_labels_reference(lay, q, g, gt_labels, matched_idx, late, n_det, n_tracks, num_classes):
    n_layers = late.shape[0]
    labels = torch.full((n_layers, n_det + n_tracks), num_classes, dtype=torch.int64, device=late.device)
    labels[lay, q] = gt_labels[g]
    if n_tracks > 0:
        has = matched_idx >= 0
        tr_lab = torch.where(has, gt_labels[matched_idx.clamp(min=0)] if gt_labels.shape[0] > 0
                             else torch.full_like(matched_idx, num_classes), torch.full_like(matched_idx, num_classes))
        labels[:, n_det:] = torch.where(late[:, None], tr_lab[None, :], labels[:, n_det:])
    return labels


# --------------------------------------------------------------------------------------------------------------
# paired L1 + GIoU box losses (models/criterion.py of the reference, :417-440)
# --------------------------------------------------------------------------------------------------------------
class _PairBoxLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, boxes_all, lay, qidx, b, tgt_boxes, gidx, weight):
        n_layers, B, Nq, _ = boxes_all.shape
        n = lay.shape[0]
        l1 = torch.empty((n,), dtype=torch.float32, device=boxes_all.device)
        gl = torch.empty_like(l1)
        L = _lib()
        L.check(L.lib.clipops_pair_box_loss_fwd_f32(
            boxes_all.data_ptr(), lay.data_ptr(), qidx.data_ptr(), B * Nq, b * Nq, tgt_boxes.data_ptr(),
            None if gidx is None else gidx.data_ptr(), None if weight is None else weight.data_ptr(), n,
            l1.data_ptr(), gl.data_ptr(), _stream(boxes_all)), "clipops_pair_box_loss_fwd_f32")
        ctx.save_for_backward(boxes_all, lay, qidx, tgt_boxes, gidx, weight)
        ctx.b = b
        return l1, gl

    @staticmethod
    def backward(ctx, g_l1, g_gl):
        boxes_all, lay, qidx, tgt_boxes, gidx, weight = ctx.saved_tensors
        n_layers, B, Nq, _ = boxes_all.shape
        grad = torch.zeros_like(boxes_all)
        L = _lib()
        L.check(L.lib.clipops_pair_box_loss_bwd_f32(
            boxes_all.data_ptr(), lay.data_ptr(), qidx.data_ptr(), B * Nq, ctx.b * Nq, tgt_boxes.data_ptr(),
            None if gidx is None else gidx.data_ptr(), None if weight is None else weight.data_ptr(), lay.shape[0],
            g_l1.contiguous().data_ptr(), g_gl.contiguous().data_ptr(), grad.data_ptr(), _stream(boxes_all)),
            "clipops_pair_box_loss_bwd_f32")
        return grad, None, None, None, None, None, None


def pair_box_loss(boxes_all: torch.Tensor, lay: torch.Tensor, qidx: torch.Tensor, b: int, tgt_boxes: torch.Tensor,
                  gidx: torch.Tensor = None, weight: torch.Tensor = None):
    """L1 (summed over the 4 coordinates) and 1 - GIoU of the pairs (boxes_all[lay[i], b, qidx[i]], tgt_boxes[gidx[i]]
    or tgt_boxes[i]), each optionally times weight[i].  boxes_all: (n_layers, B, Nq, 4) contiguous, cxcywh; the
    (lay, qidx) pairs of one call must be distinct.  Returns two (n,) tensors; gradients flow to boxes_all."""
    if not boxes_all.is_contiguous():
        boxes_all = boxes_all.contiguous()
    return _PairBoxLoss.apply(boxes_all, lay.contiguous(), qidx.contiguous(), int(b), tgt_boxes.contiguous(),
                              None if gidx is None else gidx.contiguous(),
                              None if weight is None else weight.contiguous())


def pair_box_loss_reference(boxes_all, lay, qidx, b, tgt_boxes, gidx=None, weight=None):
    from ..models.criterion import paired_giou
    from ..utils.box_ops import box_cxcywh_to_xyxy
    pb = boxes_all[lay, b, qidx]
    tb = tgt_boxes if gidx is None else tgt_boxes[gidx]
    l1 = F.l1_loss(pb, tb, reduction="none").sum(-1)
    gl = 1 - paired_giou(box_cxcywh_to_xyxy(pb), box_cxcywh_to_xyxy(tb))
    if weight is not None:
        l1, gl = l1 * weight, gl * weight
    return l1, gl


def pair_iou(boxes: torch.Tensor, tgt_boxes: torch.Tensor, gidx: torch.Tensor = None) -> torch.Tensor:
    """IoU of cxcywh boxes[i] with tgt_boxes[gidx[i]] (or tgt_boxes[i]); no gradient (the tracks' ``iou`` field only
    feeds threshold comparisons in the query updater)."""
    boxes, tgt_boxes = boxes.detach().contiguous(), tgt_boxes.contiguous()
    n = boxes.shape[0]
    out = torch.empty((n,), dtype=torch.float32, device=boxes.device)
    L = _lib()
    L.check(L.lib.clipops_pair_iou_f32(boxes.data_ptr(), tgt_boxes.data_ptr(), None if gidx is None else
                                       gidx.contiguous().data_ptr(), n, out.data_ptr(), _stream(boxes)),
            "clipops_pair_iou_f32")
    return out


def assign(cost: torch.Tensor):
    """Linear sum assignment of a batch of cost matrices (P, R, C) (any strides) on the device:
    (row_ind (P, k) int32, col_ind (P, k) int32, status (P,) int32), k = min(R, C) -- per problem the pairs
    ``scipy.optimize.linear_sum_assignment`` returns (same algorithm, scan order and tie rule:
    csrc/assign_core.h), status k, or -1 where scipy would raise "cost matrix is infeasible".  No synchronisation."""
    if cost.dim() != 3 or cost.dtype != torch.float32 or not cost.is_cuda:
        raise RuntimeError("assign expects a float32 CUDA tensor of shape (problems, rows, cols)")
    P, R, C = cost.shape
    k = min(R, C)
    row = torch.empty((P, k), dtype=torch.int32, device=cost.device)
    col = torch.empty((P, k), dtype=torch.int32, device=cost.device)
    status = torch.empty((P,), dtype=torch.int32, device=cost.device)
    L = _lib()
    L.check(L.lib.clipops_assign_f32(cost.data_ptr(), cost.stride(0), cost.stride(1), cost.stride(2), P, R, C,
                                     row.data_ptr(), col.data_ptr(), status.data_ptr(), _stream(cost)),
            "clipops_assign_f32")
    return row, col, status


def pair_iou_reference(boxes, tgt_boxes, gidx=None):
    from ..models.criterion import paired_iou
    from ..utils.box_ops import box_cxcywh_to_xyxy
    tb = tgt_boxes if gidx is None else tgt_boxes[gidx]
    return paired_iou(box_cxcywh_to_xyxy(boxes), box_cxcywh_to_xyxy(tb))


# --------------------------------------------------------------------------------------------------------------
# focal loss of stacked layers (models/criterion.py of the reference, :442-467)
# --------------------------------------------------------------------------------------------------------------
class _FocalPerLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, alpha, gamma):
        n_layers, Nq, K = logits.shape
        sl, sq = _strides3(logits)
        loss = torch.empty((n_layers,), dtype=torch.float32, device=logits.device)
        L = _lib()
        L.check(L.lib.clipops_focal_fwd_f32(logits.data_ptr(), sl, sq, labels.data_ptr(), n_layers, Nq, K, alpha, gamma,
                                            loss.data_ptr(), _stream(logits)), "clipops_focal_fwd_f32")
        ctx.save_for_backward(logits, labels)
        ctx.alpha, ctx.gamma = alpha, gamma
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, labels = ctx.saved_tensors
        n_layers, Nq, K = logits.shape
        sl, sq = _strides3(logits)
        grad = torch.empty((n_layers, Nq, K), dtype=torch.float32, device=logits.device)
        L = _lib()
        L.check(L.lib.clipops_focal_bwd_f32(logits.data_ptr(), sl, sq, labels.data_ptr(), n_layers, Nq, K, ctx.alpha,
                                            ctx.gamma, g.contiguous().data_ptr(), grad.data_ptr(), _stream(logits)),
                "clipops_focal_bwd_f32")
        return grad, None, None, None


def focal_loss_per_layer(logits: torch.Tensor, labels: torch.Tensor, alpha: float = 0.25, gamma: float = 2.0):
    """(n_layers, Nq, K) logits (a view is fine), (n_layers, Nq) int64 labels with K = background -> (n_layers,)
    sigmoid focal loss, mean over classes and sum over queries."""
    return _FocalPerLayer.apply(logits, labels.contiguous(), float(alpha), float(gamma))


def focal_loss_per_layer_reference(logits, labels, alpha: float = 0.25, gamma: float = 2.0):
    from ..models.criterion import sigmoid_focal_loss_per_layer
    K = logits.shape[-1]
    one_hot = F.one_hot(labels, K + 1)[..., :-1].to(logits.dtype)
    return sigmoid_focal_loss_per_layer(logits, one_hot, alpha, gamma)


# --------------------------------------------------------------------------------------------------------------
# decoder glue: sine embedding of the anchors, clamped logit, box refinement
# --------------------------------------------------------------------------------------------------------------
class _SineEmbed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, dim_t, scale):
        K, F_ = pos.shape[-1], dim_t.shape[0]
        n = pos.numel() // K
        out = torch.empty(pos.shape[:-1] + (K * F_,), dtype=torch.float32, device=pos.device)
        L = _lib()
        L.check(L.lib.clipops_sine_embed_fwd_f32(pos.data_ptr(), dim_t.data_ptr(), n, K, F_, scale, out.data_ptr(),
                                                 _stream(pos)), "clipops_sine_embed_fwd_f32")
        ctx.save_for_backward(pos, dim_t)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, g):
        pos, dim_t = ctx.saved_tensors
        K, F_ = pos.shape[-1], dim_t.shape[0]
        grad = torch.empty_like(pos)
        L = _lib()
        L.check(L.lib.clipops_sine_embed_bwd_f32(pos.data_ptr(), dim_t.data_ptr(), pos.numel() // K, K, F_, ctx.scale,
                                                 g.contiguous().data_ptr(), grad.data_ptr(), _stream(pos)),
                "clipops_sine_embed_bwd_f32")
        return grad, None, None


def sine_embed(pos: torch.Tensor, dim_t: torch.Tensor, scale: float) -> torch.Tensor:
    """(..., K) -> (..., K * F): out[..., k*F + j] = (sin | cos)(pos[..., k] * scale / dim_t[j]), even j sin, odd j cos."""
    return _SineEmbed.apply(pos.contiguous(), dim_t, float(scale))


class _InverseSigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        y = torch.empty_like(x)
        L = _lib()
        L.check(L.lib.clipops_inverse_sigmoid_fwd_f32(x.data_ptr(), x.numel(), eps, y.data_ptr(), _stream(x)),
                "clipops_inverse_sigmoid_fwd_f32")
        ctx.save_for_backward(x)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        gx = torch.empty_like(x)
        L = _lib()
        L.check(L.lib.clipops_inverse_sigmoid_bwd_f32(x.data_ptr(), g.contiguous().data_ptr(), x.numel(), ctx.eps,
                                                      gx.data_ptr(), _stream(x)), "clipops_inverse_sigmoid_bwd_f32")
        return gx, None


def inverse_sigmoid(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    return _InverseSigmoid.apply(x.contiguous(), float(eps))


class _RefineBoxes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, delta, ref, eps):
        out = torch.empty_like(delta)
        L = _lib()
        L.check(L.lib.clipops_refine_boxes_fwd_f32(delta.data_ptr(), ref.data_ptr(), delta.numel(), eps, out.data_ptr(),
                                                   _stream(delta)), "clipops_refine_boxes_fwd_f32")
        ctx.save_for_backward(out, ref)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, g):
        out, ref = ctx.saved_tensors
        gd = torch.empty_like(out)
        gr = torch.empty_like(out) if ctx.needs_input_grad[1] else None
        L = _lib()
        L.check(L.lib.clipops_refine_boxes_bwd_f32(out.data_ptr(), ref.data_ptr(), g.contiguous().data_ptr(),
                                                   out.numel(), ctx.eps, gd.data_ptr(),
                                                   None if gr is None else gr.data_ptr(), _stream(out)),
                "clipops_refine_boxes_bwd_f32")
        return gd, gr, None


def refine_boxes(delta: torch.Tensor, ref: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """sigmoid(delta + inverse_sigmoid(ref)) for same-shaped tensors."""
    if delta.shape != ref.shape:
        raise RuntimeError("refine_boxes: delta and ref must have the same shape")
    return _RefineBoxes.apply(delta.contiguous(), ref.contiguous(), float(eps))


# --------------------------------------------------------------------------------------------------------------
# bias gradients of the query-sized linears
# --------------------------------------------------------------------------------------------------------------
COLSUM_MAX_ROWS = 2048          # one pass of the tile kernel
COLSUM_CHUNK_ROWS = 256         # tall matrices: partial sums per 256 rows, then one pass over the partials


def colsum(x: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """x.sum(0) of a contiguous 2-d CUDA matrix through the tiled kernels (fixed summation order, fp32 sums and result):
    fp32 input -- one pass up to COLSUM_MAX_ROWS rows, two passes (per-chunk partial sums, then the partials) above;
    bf16 input -- always the two-pass form.  Anything else takes torch's reduction.  ``out`` may be a contiguous
    fp32 (cols,) destination."""
    b16 = x.dtype == torch.bfloat16
    ok = (x.dim() == 2 and x.is_contiguous() and x.is_cuda and (x.dtype == torch.float32 or b16)
          and os.environ.get("MEMOTR_FUSED_CLIP_OPS", "1") != "0"
          and (out is None or (out.is_contiguous() and out.dtype == torch.float32))
          and 0 < x.shape[0] <= COLSUM_MAX_ROWS * COLSUM_CHUNK_ROWS)
    if not ok:
        return torch.sum(x, 0, out=out) if out is not None else x.sum(0)
    if x.shape[0] > COLSUM_MAX_ROWS and not b16 and os.environ.get("MEMOTR_COLSUM_TALL", "1") == "0":
        return torch.sum(x, 0, out=out) if out is not None else x.sum(0)
    L = _lib()
    if x.shape[0] > COLSUM_MAX_ROWS or b16:
        chunks = -(-x.shape[0] // COLSUM_CHUNK_ROWS)
        partial = torch.empty((chunks, x.shape[1]), dtype=torch.float32, device=x.device)
        fn = L.lib.clipops_colsum_partial_bf16 if b16 else L.lib.clipops_colsum_partial_f32
        L.check(fn(x.data_ptr(), x.shape[0], x.shape[1], COLSUM_CHUNK_ROWS, partial.data_ptr(), _stream(x)),
                "clipops_colsum_partial")
        x = partial
    if out is None:
        out = torch.empty((x.shape[1],), dtype=torch.float32, device=x.device)
    L.check(L.lib.clipops_colsum_f32(x.data_ptr(), x.shape[0], x.shape[1], out.data_ptr(), _stream(x)),
            "clipops_colsum_f32")
    return out


def relu_bwd_colsum_usable(g: torch.Tensor, y: torch.Tensor) -> bool:
    return (os.environ.get("MEMOTR_FUSED_CLIP_OPS", "1") != "0" and os.environ.get("MEMOTR_FUSED_RELU_COLSUM", "1") != "0"
            and g.is_cuda and g.dim() == 2 and g.is_contiguous() and y.is_contiguous() and y.shape == g.shape
            and g.dtype == y.dtype and g.dtype == torch.float32        # (bf16: 546 vs 358 us for the two torch passes)
            and COLSUM_MAX_ROWS < g.shape[0] <= COLSUM_MAX_ROWS * COLSUM_CHUNK_ROWS)


def relu_bwd_colsum(g: torch.Tensor, y: torch.Tensor):
    """(g * (y > 0), its column sums in fp32) for tall contiguous (rows, cols) matrices: the ReLU mask of a fused-ReLU
    Linear's backward and its bias gradient with ONE pass over the gradient instead of two."""
    rows, cols = g.shape
    chunks = -(-rows // COLSUM_CHUNK_ROWS)
    g2 = torch.empty_like(g)
    partial = torch.empty((chunks, cols), dtype=torch.float32, device=g.device)
    out = torch.empty((cols,), dtype=torch.float32, device=g.device)
    L = _lib()
    fn = L.lib.clipops_relu_bwd_colsum_partial_f32
    L.check(fn(g.data_ptr(), y.data_ptr(), rows, cols, COLSUM_CHUNK_ROWS, g2.data_ptr(), partial.data_ptr(), _stream(g)),
            "clipops_relu_bwd_colsum_partial")
    L.check(L.lib.clipops_colsum_f32(partial.data_ptr(), chunks, cols, out.data_ptr(), _stream(g)), "clipops_colsum_f32")
    return g2, out


class _AddRowBias(torch.autograd.Function):
    """x (..., C) + bias (C,): the bias gradient through the tiled column sums instead of torch's reduction."""

    @staticmethod
    def forward(ctx, x, bias):
        return x + bias

    @staticmethod
    def backward(ctx, g):
        gb = None
        if ctx.needs_input_grad[1]:
            g2 = g.reshape(-1, g.shape[-1])
            gb = colsum(g2 if g2.is_contiguous() else g2.contiguous())
        return (g if ctx.needs_input_grad[0] else None), gb


def add_row_bias(x: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    if x.is_cuda and bias.requires_grad and torch.is_grad_enabled() and bias.dim() == 1:
        return _AddRowBias.apply(x, bias)
    return x + bias


# --------------------------------------------------------------------------------------------------------------
# backward of a query-sized Linear in one launch (include/clip_ops_hip.h: clipops_linear_bwd_f32)
# --------------------------------------------------------------------------------------------------------------
LINEAR_BWD_MAX_ROWS = 1024          # beyond a few hundred rows the library GEMMs win
LINEAR_BWD_MAX_OUT = 512            # ... and for wide outputs too: 32 x 32 tiles re-read their operands from L2 (measured
                                    # inside a graph at 320 rows, tools/linear_bwd_probe.py: 256 -> 256: 6.5 vs 14.5 us for
                                    # the chain, with ReLU 10.5 vs 18.4; 256 -> 768: 12.7 vs 11.5; 256 -> 1024: 18.4 vs 17.9)


def linear_bwd_usable(g2: torch.Tensor, x2: torch.Tensor, weight: torch.Tensor) -> bool:
    return (os.environ.get("MEMOTR_FUSED_LINEAR_BWD", "1") != "0" and fused(g2, x2, weight)
            and g2.dtype == torch.float32 and x2.dtype == torch.float32 and weight.dtype == torch.float32
            and 0 < g2.shape[0] <= LINEAR_BWD_MAX_ROWS and g2.shape[1] <= LINEAR_BWD_MAX_OUT
            and x2.is_contiguous() and weight.is_contiguous())


def linear_bwd(g2: torch.Tensor, y_relu, x2: torch.Tensor, weight: torch.Tensor, need_x=True, need_w=True, need_b=True,
               gw_out: torch.Tensor = None, gb_out: torch.Tensor = None):
    """(grad_x, grad_w, grad_b) of y = x2 @ weight.T + b for g2 = d/dy (rows, out); ``y_relu``: the forward's output
    when it applied a ReLU (the mask y > 0 is then part of the kernel).  ``gw_out`` / ``gb_out``: contiguous buffers
    to write into (row slices of a packed parameter's gradient)."""
    g2 = g2.contiguous()
    rows, out_f = g2.shape
    in_f = x2.shape[1]
    need_w = need_w or need_b
    gx = torch.empty((rows, in_f), dtype=torch.float32, device=g2.device) if need_x else None
    gw = (gw_out if gw_out is not None else torch.empty((out_f, in_f), dtype=torch.float32, device=g2.device)) if need_w else None
    gb = (gb_out if gb_out is not None else torch.empty((out_f,), dtype=torch.float32, device=g2.device)) if need_b else None
    assert gw is None or (gw.is_contiguous() and gw.shape == (out_f, in_f)), "gw_out must be a contiguous (out, in) buffer"
    assert gb is None or (gb.is_contiguous() and gb.shape == (out_f,)), "gb_out must be a contiguous (out,) buffer"
    L_ = _lib()
    L_.check(L_.lib.clipops_linear_bwd_f32(g2.data_ptr(), None if y_relu is None else y_relu.data_ptr(), x2.data_ptr(),
                                           weight.data_ptr(), rows, in_f, out_f,
                                           None if gx is None else gx.data_ptr(), None if gw is None else gw.data_ptr(),
                                           None if gb is None else gb.data_ptr(), _stream(g2)), "clipops_linear_bwd_f32")
    return gx, gw, gb


LINEAR_FWD_MAX_IN = 256             # (inside a graph, 320 rows: 256 -> 256 5.0 vs 5.4 us for the library, 512 -> 256 7.4 vs 6.0)


def linear_fwd_usable(x2: torch.Tensor, weight: torch.Tensor, bias) -> bool:
    return (os.environ.get("MEMOTR_FUSED_LINEAR_FWD", "1") != "0" and fused(x2, weight)
            and x2.dtype == torch.float32 and weight.dtype == torch.float32
            and 0 < x2.shape[0] <= LINEAR_BWD_MAX_ROWS and weight.shape[0] <= LINEAR_BWD_MAX_OUT
            and x2.shape[1] % 4 == 0 and x2.shape[1] <= LINEAR_FWD_MAX_IN and x2.is_contiguous() and weight.is_contiguous()
            and (bias is None or (bias.dtype == torch.float32 and bias.is_contiguous())))


def linear_fwd(x2: torch.Tensor, weight: torch.Tensor, bias, relu: bool) -> torch.Tensor:
    """[relu](x2 @ weight.T + bias) for a few hundred rows: one 5 us launch where the library GEMM takes 7-8."""
    rows, in_f = x2.shape
    out_f = weight.shape[0]
    y = torch.empty((rows, out_f), dtype=torch.float32, device=x2.device)
    L_ = _lib()
    L_.check(L_.lib.clipops_linear_fwd_f32(x2.data_ptr(), weight.data_ptr(), None if bias is None else bias.data_ptr(),
                                           rows, in_f, out_f, 1 if relu else 0, y.data_ptr(), _stream(x2)),
             "clipops_linear_fwd_f32")
    return y


def linear_bwd_reference(g2, y_relu, x2, weight):
    if y_relu is not None:
        g2 = g2 * (y_relu > 0).to(g2.dtype)
    return g2 @ weight, g2.t() @ x2, g2.sum(0)


# --------------------------------------------------------------------------------------------------------------
# the backbone's element-wise tail: frozen-BN shift (+ residual) + ReLU in one pass (include/clip_ops_hip.h, ABI 9)
# --------------------------------------------------------------------------------------------------------------
class _ShiftRelu(torch.autograd.Function):
    """y = relu(x + shift[None, :, None, None] (+ res)), written over ``x`` (a convolution's fresh output)."""

    @staticmethod
    def forward(ctx, x, shift, res):
        N, C, H, W = x.shape
        L_ = _lib()
        fn = L_.lib.clipops_shift_relu_bf16 if x.dtype == torch.bfloat16 else L_.lib.clipops_shift_relu_f32
        L_.check(fn(x.data_ptr(), shift.data_ptr(), None if res is None else res.data_ptr(), N * C, C, H * W,
                    _stream(x)), "clipops_shift_relu")
        ctx.mark_dirty(x)
        ctx.save_for_backward(x)
        ctx.has_res = res is not None
        return x

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        gx = torch.ops.aten.threshold_backward(g, y, 0)      # (the ReLU's own backward kernel; the shift is a constant)
        return gx, None, (gx if ctx.has_res else None)


def shift_relu_reference(x: torch.Tensor, shift: torch.Tensor, res: torch.Tensor = None) -> torch.Tensor:
    y = x + shift.to(x.dtype)[None, :, None, None]
    if res is not None:
        y = y + res
    return F.relu(y)


def shift_relu_(x: torch.Tensor, shift: torch.Tensor, res: torch.Tensor = None) -> torch.Tensor:
    """relu(x + per-channel shift (+ res)) over an NCHW tensor; in place on CUDA fp32 / bf16 (``x`` must be a tensor
    nothing else reads: the output of the convolution in front), the element-wise torch chain elsewhere."""
    ok = (os.environ.get("MEMOTR_FUSED_CLIP_OPS", "1") != "0" and os.environ.get("MEMOTR_FUSED_SHIFT_RELU", "1") != "0"
          and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous()
          and not x._is_view() and shift.dtype == torch.float32 and shift.is_contiguous() and not shift.requires_grad
          and (res is None or (res.dtype == x.dtype and res.shape == x.shape and res.is_contiguous())))
    if not ok:
        return shift_relu_reference(x, shift, res)
    return _ShiftRelu.apply(x, shift, res)


# --------------------------------------------------------------------------------------------------------------
# self-attention over the decoder queries (head_dim 32)
# --------------------------------------------------------------------------------------------------------------
MHA_MAX_L = 512
MHA_HEAD_DIM = 32


class _SelfAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qk_p, v_p, key_mask, n_heads):
        B, L, E2 = qk_p.shape
        E = E2 // 2
        out = torch.empty((B, L, E), dtype=torch.float32, device=qk_p.device)
        lse = torch.empty((B, n_heads, L), dtype=torch.float32, device=qk_p.device)
        scale = 1.0 / (E // n_heads) ** 0.5
        mk = None if key_mask is None else key_mask.data_ptr()
        L_ = _lib()
        L_.check(L_.lib.clipops_mha_fwd_f32(qk_p.data_ptr(), qk_p.data_ptr() + 4 * E, v_p.data_ptr(), L * E2, E2, L * E2,
                                            E2, L * E, E, mk, B, n_heads, L, scale, out.data_ptr(), lse.data_ptr(),
                                            _stream(qk_p)), "clipops_mha_fwd_f32")
        ctx.save_for_backward(qk_p, v_p, key_mask, out, lse)
        ctx.n_heads, ctx.scale = n_heads, scale
        return out

    @staticmethod
    def backward(ctx, g):
        qk_p, v_p, key_mask, out, lse = ctx.saved_tensors
        B, L, E2 = qk_p.shape
        E = E2 // 2
        g = g.contiguous()
        g_qk, g_v = torch.empty_like(qk_p), torch.empty_like(v_p)
        mk = None if key_mask is None else key_mask.data_ptr()
        L_ = _lib()
        L_.check(L_.lib.clipops_mha_bwd_f32(qk_p.data_ptr(), qk_p.data_ptr() + 4 * E, v_p.data_ptr(), L * E2, E2, L * E2,
                                            E2, L * E, E, mk, out.data_ptr(), lse.data_ptr(), g.data_ptr(), B, ctx.n_heads,
                                            L, ctx.scale, g_qk.data_ptr(), L * E2, E2, g_qk.data_ptr() + 4 * E, L * E2,
                                            E2, g_v.data_ptr(), L * E, E, _stream(qk_p)), "clipops_mha_bwd_f32")
        return g_qk, g_v, None, None


class _Attention3(torch.autograd.Function):
    """The same kernels for separately projected q, k, v (B, L, E) -- the query updater's memory attention, whose
    queries and keys come from different inputs.  ``key_mask`` (B, L) bool, True = ignore that key (the padded slots of
    the captured update, models/updater_graphs.py), or None."""

    @staticmethod
    def forward(ctx, q_p, k_p, v_p, key_mask, n_heads):
        B, L, E = q_p.shape
        out = torch.empty((B, L, E), dtype=torch.float32, device=q_p.device)
        lse = torch.empty((B, n_heads, L), dtype=torch.float32, device=q_p.device)
        scale = 1.0 / (E // n_heads) ** 0.5
        mk = None if key_mask is None else key_mask.data_ptr()
        L_ = _lib()
        L_.check(L_.lib.clipops_mha_fwd_f32(q_p.data_ptr(), k_p.data_ptr(), v_p.data_ptr(), L * E, E, L * E, E, L * E, E,
                                            mk, B, n_heads, L, scale, out.data_ptr(), lse.data_ptr(), _stream(q_p)),
                 "clipops_mha_fwd_f32")
        ctx.save_for_backward(q_p, k_p, v_p, key_mask, out, lse)
        ctx.n_heads, ctx.scale = n_heads, scale
        return out

    @staticmethod
    def backward(ctx, g):
        q_p, k_p, v_p, key_mask, out, lse = ctx.saved_tensors
        B, L, E = q_p.shape
        g = g.contiguous()
        g_q, g_k, g_v = torch.empty_like(q_p), torch.empty_like(k_p), torch.empty_like(v_p)
        mk = None if key_mask is None else key_mask.data_ptr()
        L_ = _lib()
        L_.check(L_.lib.clipops_mha_bwd_f32(q_p.data_ptr(), k_p.data_ptr(), v_p.data_ptr(), L * E, E, L * E, E, L * E, E,
                                            mk, out.data_ptr(), lse.data_ptr(), g.data_ptr(), B, ctx.n_heads, L,
                                            ctx.scale, g_q.data_ptr(), L * E, E, g_k.data_ptr(), L * E, E,
                                            g_v.data_ptr(), L * E, E, _stream(q_p)), "clipops_mha_bwd_f32")
        return g_q, g_k, g_v, None, None


def attention(q_p: torch.Tensor, k_p: torch.Tensor, v_p: torch.Tensor, n_heads: int, key_padding_mask=None) -> torch.Tensor:
    """softmax(q k^T / sqrt(d)) v per head for projected (B, L, E) inputs of one common length; (B, L, E) out.
    ``key_padding_mask`` (B, L) bool: True = that key takes no part."""
    mask = None if key_padding_mask is None else key_padding_mask.contiguous()
    return _Attention3.apply(q_p.contiguous(), k_p.contiguous(), v_p.contiguous(), mask, int(n_heads))


def attention_supported(q_p: torch.Tensor, k_p: torch.Tensor, n_heads: int) -> bool:
    return (os.environ.get("MEMOTR_ATTN_KERNELS", "1") != "0" and q_p.dim() == 3 and q_p.shape == k_p.shape
            and fused(q_p, k_p) and q_p.dtype == torch.float32
            and q_p.shape[2] % n_heads == 0 and q_p.shape[2] // n_heads == MHA_HEAD_DIM and 0 < q_p.shape[1] <= MHA_MAX_L)


def self_attention_supported(qk_p: torch.Tensor, n_heads: int, any_float: bool = False) -> bool:
    """``any_float``: the caller casts a bf16 / fp16 projection to float32 first (autocast island)."""
    B, L, E2 = qk_p.shape
    dtype_ok = qk_p.is_floating_point() if any_float else qk_p.dtype == torch.float32
    return (os.environ.get("MEMOTR_ATTN_KERNELS", "1") != "0" and os.environ.get("MEMOTR_FUSED_CLIP_OPS", "1") != "0"
            and qk_p.is_cuda and dtype_ok and E2 % (2 * n_heads) == 0
            and E2 // 2 // n_heads == MHA_HEAD_DIM and 0 < L <= MHA_MAX_L)


def self_attention(qk_p: torch.Tensor, v_p: torch.Tensor, key_padding_mask, n_heads: int) -> torch.Tensor:
    """softmax(q k^T / sqrt(d)) v per head from the packed projections: qk_p (B, L, 2E) = [q | k], v_p (B, L, E);
    key_padding_mask (B, L) bool (True = ignore that key) or None.  Returns (B, L, E), heads concatenated -- the
    layout the output projection takes, no transposes or copies either side."""
    mask = None if key_padding_mask is None else key_padding_mask.contiguous()
    return _SelfAttention.apply(qk_p.contiguous(), v_p.contiguous(), mask, int(n_heads))


def self_attention_reference(qk_p, v_p, key_padding_mask, n_heads):
    B, L, E2 = qk_p.shape
    E, d = E2 // 2, E2 // 2 // n_heads
    q, k = (t.transpose(1, 2) for t in qk_p.view(B, L, 2, n_heads, d).unbind(2))
    v = v_p.view(B, L, n_heads, d).transpose(1, 2)
    att = (q @ k.transpose(-1, -2)) / d ** 0.5
    if key_padding_mask is not None:
        att = att.masked_fill(key_padding_mask.view(B, 1, 1, L), float("-inf"))
    return (att.softmax(-1) @ v).transpose(1, 2).reshape(B, L, E)


# --------------------------------------------------------------------------------------------------------------
# residual add + LayerNorm (rows of 256)
# --------------------------------------------------------------------------------------------------------------
LN_COLS = 256


class _AddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, weight, bias, eps):
        rows = x.numel() // LN_COLS
        s = torch.empty_like(x)
        y = torch.empty_like(x)
        stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
        L = _lib()
        L.check(L.lib.clipops_add_layer_norm_fwd_f32(x.data_ptr(), res.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                                     rows, eps, s.data_ptr(), y.data_ptr(), stats.data_ptr(), _stream(x)),
                "clipops_add_layer_norm_fwd_f32")
        ctx.save_for_backward(s, stats, weight)
        return y

    @staticmethod
    def backward(ctx, g):
        s, stats, weight = ctx.saved_tensors
        rows = s.numel() // LN_COLS
        g = g.contiguous()
        gs = torch.empty_like(s)
        chunk = 16 if rows <= 4096 else 64
        partial = torch.empty((-(-rows // chunk), 2 * LN_COLS), dtype=torch.float32, device=s.device)
        L = _lib()
        L.check(L.lib.clipops_add_layer_norm_bwd_f32(g.data_ptr(), s.data_ptr(), stats.data_ptr(), weight.data_ptr(), rows,
                                                     chunk, gs.data_ptr(), partial.data_ptr(), _stream(s)),
                "clipops_add_layer_norm_bwd_f32")
        gwb = colsum(partial)                       # [grad_gamma | grad_beta]
        return gs, gs, gwb[:LN_COLS], gwb[LN_COLS:], None


def add_layer_norm_supported(x: torch.Tensor, res: torch.Tensor, norm) -> bool:
    """fp32 CUDA tensors; under autocast also bf16 / fp16 ones (see add_layer_norm)."""
    island = torch.is_autocast_enabled()
    ok_dtype = (lambda t: t.is_floating_point()) if island else (lambda t: t.dtype == torch.float32)
    return (isinstance(norm, torch.nn.LayerNorm) and norm.elementwise_affine and norm.bias is not None
            and tuple(norm.normalized_shape) == (LN_COLS,) and x.shape == res.shape and x.shape[-1] == LN_COLS
            and x.is_cuda and res.is_cuda and ok_dtype(x) and ok_dtype(res) and norm.weight.dtype == torch.float32
            and os.environ.get("MEMOTR_FUSED_CLIP_OPS", "1") != "0" and os.environ.get("MEMOTR_FUSED_LN", "1") != "0")


def add_layer_norm(x: torch.Tensor, res: torch.Tensor, norm) -> torch.Tensor:
    """norm(x + res) for an nn.LayerNorm over 256 features: one kernel forward (sum, statistics and output in one
    pass), one + a column sum backward; other shapes / dtypes / devices go through the module.

    Under autocast the kernel runs as an fp32 island: autocast itself evaluates layer_norm in float32 (and the sum
    ``x + res`` promotes to float32 as soon as one operand is), so casting the low-precision operand up front gives the
    values autocast would have produced -- through one kernel instead of add + native_layer_norm (+ their backward)."""
    if add_layer_norm_supported(x, res, norm):
        return _AddLayerNorm.apply(x.float().contiguous(), res.float().contiguous(), norm.weight, norm.bias,
                                   float(norm.eps))
    return norm(x + res)
