"""Linear layers over the flattened pyramid (tens of thousands of rows).

``y = x W^T + b`` with ``x`` of shape (..., K) and M = prod(...) rows.  Forward and input gradient are ordinary
GEMMs.  The weight gradient ``dW = dY^T X`` contracts over M (22,323 at 800x1333) into a small (N, K) output:
a single GEMM call leaves most of the 256 CUs idle (measured on MI355X, fp32, hipBLASLt: 27 TFLOP/s for
N = K = 256, 69 for 2048x256).  Splitting the contraction into row chunks -- one batched GEMM + a sum of the
partial products -- keeps the chip busy: 64 / 111 TFLOP/s for the same shapes (tools/gemm_probe.py).
Only the summation order of the fp32 partial products changes (relative difference ~3e-6).
"""
from __future__ import annotations

import contextlib
import os

import torch
import torch.nn.functional as F

MIN_ROWS = 8192          # below this the plain GEMM is already fine (decoder-sized inputs)
TARGET_CHUNKS = 24
ROCBLAS_DGRAD_ROWS = 65536   # input-gradient GEMMs with at least this many rows go to rocBLAS (clip-batched encoder)


@contextlib.contextmanager
def prefer_blas(lib: str):
    """Route ``mm`` / ``bmm`` to "cublas" (= rocBLAS) or "cublaslt" (= hipBLASLt) inside the block."""
    prev = torch.backends.cuda.preferred_blas_library()
    if prev == _BACKENDS[lib]:
        yield
        return
    torch.backends.cuda.preferred_blas_library(lib)
    try:
        yield
    finally:
        torch.backends.cuda.preferred_blas_library(prev)


_BACKENDS = {"cublas": torch._C._BlasBackend.Cublas, "cublaslt": torch._C._BlasBackend.Cublaslt}


def configure_blas() -> str:
    """Process-wide GEMM library choice for ``mm`` / ``bmm`` (``addmm`` with a bias keeps hipBLASLt's fused epilogue
    either way).  Measured on MI355X, fp32 (tools/small_gemm_probe.py, tools/gemm_probe.py):

    * decoder-sized GEMMs (a few hundred rows): hipBLASLt costs ~19 us of HOST time per call and maps the
      (256 x K) x (K x 256) weight gradient to a single 256x256 macro-tile -- 80 us for 40 MFLOP, 290 times per
      step; rocBLAS: ~7 us host, 7-8 us GPU for the same calls;
    * pyramid-sized GEMMs: the two libraries are within a few percent of each other, except the 22,323-row input
      gradients (hipBLASLt 33 vs rocBLAS 68 us) -- ``long_linear`` picks per call.

    So rocBLAS is the default and ``long_linear`` opts back into hipBLASLt where it wins.
    MEMOTR_BLAS=hipblaslt|rocblas|keep overrides.  Returns the choice."""
    choice = os.environ.get("MEMOTR_BLAS", "rocblas")
    if choice == "rocblas":
        torch.backends.cuda.preferred_blas_library("cublas")
    elif choice == "hipblaslt":
        torch.backends.cuda.preferred_blas_library("cublaslt")
    return choice


def _pick_chunks(rows: int) -> int:
    for c in range(TARGET_CHUNKS, TARGET_CHUNKS + 24):      # prefer an exact divisor near the target
        if rows % c == 0:
            return c
    for c in range(TARGET_CHUNKS - 1, 11, -1):
        if rows % c == 0:
            return c
    return TARGET_CHUNKS


# bias + ReLU in the GEMM epilogue (hipBLASLt RELU_BIAS) for the encoder FFN's first linear: the activation is 8x the
# model width, and a separate in-place ReLU re-reads and re-writes it (22,323 x 2048 x frames floats per layer).
FUSE_RELU_EPILOGUE = os.environ.get("MEMOTR_FUSE_RELU", "1") != "0"


AMP_SPLITK = os.environ.get("MEMOTR_AMP_SPLITK", "1") != "0"      # autocast: long linears keep the split-K node
AMP_MIN_ROWS = 256       # ... from this many rows on (below, torch's bias-gradient reduction is a single block)


class _SplitKLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu=False):
        ctx.has_bias = bias is not None
        ctx.relu = bool(relu)
        xc, wc, bc = x, weight, bias
        if torch.is_autocast_enabled() and x.is_cuda:     # explicit casts, then autocast off: one cast per operand
            dt = torch.get_autocast_dtype("cuda")
            xc, wc = x.to(dt), weight.to(dt)
            bc = None if bias is None else bias.to(dt)
        with torch.autocast("cuda", enabled=False) if x.is_cuda else contextlib.nullcontext():
            if ctx.relu:
                y = torch._addmm_activation(bc, xc, wc.t(), use_gelu=False)      # relu(x W^T + b), one kernel
                ctx.save_for_backward(x, weight, y)
                return y
            ctx.save_for_backward(x, weight)
            return F.linear(xc, wc, bc)

    @staticmethod
    def backward(ctx, grad_out):
        gb_done = None
        if ctx.relu:
            x, weight, y = ctx.saved_tensors
            from ..functions import clip_ops
            g2d, y2d = grad_out.reshape(-1, weight.shape[0]), y.reshape(-1, weight.shape[0])
            if ctx.has_bias and ctx.needs_input_grad[2] and clip_ops.relu_bwd_colsum_usable(g2d, y2d):
                grad_out, gb_done = clip_ops.relu_bwd_colsum(g2d, y2d)             # mask + bias-gradient partials: one pass
            else:
                grad_out = torch.ops.aten.threshold_backward(grad_out, y, 0.0)     # the ReLU mask, one pass
        else:
            x, weight = ctx.saved_tensors
        gx = gw = gb = None
        K, N = weight.shape[1], weight.shape[0]
        g2 = grad_out.reshape(-1, N)
        cdt = g2.dtype                      # bf16 under autocast (forward ran in bf16), else the parameter dtype
        on_gpu = g2.is_cuda
        if ctx.needs_input_grad[0]:
            lib = "cublas" if g2.shape[0] >= ROCBLAS_DGRAD_ROWS else "cublaslt"
            with (prefer_blas(lib) if on_gpu else contextlib.nullcontext()):
                gx = (g2 @ weight.to(cdt)).view(x.shape).to(x.dtype)
        if ctx.needs_input_grad[1]:
            x2 = x.reshape(-1, K).to(cdt)
            rows = x2.shape[0]
            c = _pick_chunks(rows)
            r = rows // c
            main = r * c
            with (prefer_blas("cublaslt") if on_gpu else contextlib.nullcontext()):
                gw = torch.bmm(g2[:main].view(c, r, N).transpose(1, 2), x2[:main].view(c, r, K)).sum(0)
                if main < rows:
                    gw = gw + g2[main:].t() @ x2[main:]
            gw = gw.to(weight.dtype)
        if gb_done is not None:
            gb = gb_done.to(weight.dtype)
        elif ctx.has_bias and ctx.needs_input_grad[2]:
            from ..functions import clip_ops
            # tiled two-pass column sum for fp32 (14 vs 33 us at 66,969 x 256); torch's reduction otherwise
            gb = (clip_ops.colsum(g2) if g2.is_contiguous() else g2.sum(0)).to(weight.dtype)
        return gx, gw, gb, None


class _RowLinear(torch.autograd.Function):
    """F.linear (optionally + ReLU in the GEMM epilogue) over a few hundred rows -- the decoder's query-sized linears --
    as one autograd node whose bias gradient goes through the tiled column-sum kernel: torch's generic reduction takes
    12-17 us for 310 x 256..2048, ~400 times per train step (tools/reduce_prof.py)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        from ..functions import clip_ops
        x2 = x.reshape(-1, x.shape[-1])
        ctx.relu = bool(relu)
        if clip_ops.linear_fwd_usable(x2, weight, bias):
            y = clip_ops.linear_fwd(x2, weight, bias, ctx.relu)                    # one MFMA launch (clip_ops ABI 9)
            ctx.save_for_backward(*((x2, weight, y) if ctx.relu else (x2, weight)))
        elif ctx.relu:
            y = torch._addmm_activation(bias, x2, weight.t(), use_gelu=False)      # relu(x W^T + b), one kernel
            ctx.save_for_backward(x2, weight, y)
        else:
            y = torch.addmm(bias, x2, weight.t())
            ctx.save_for_backward(x2, weight)
        ctx.x_shape = x.shape
        return y if x.dim() == 2 else y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, grad_out):
        from ..functions import clip_ops
        g2 = grad_out.reshape(-1, grad_out.shape[-1])
        saved = ctx.saved_tensors            # (read ONCE: under activation checkpointing a second unpack is an error)
        x2, weight, y = saved[0], saved[1], (saved[2] if ctx.relu else None)
        if clip_ops.linear_bwd_usable(g2, x2, weight):
            # [ReLU mask,] grad_x, grad_w and grad_b in ONE launch (include/clip_ops_hip.h: clipops_linear_bwd_f32)
            gx, gw, gb = clip_ops.linear_bwd(g2, y, x2, weight, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                             ctx.needs_input_grad[2])
            return (None if gx is None else gx.view(ctx.x_shape)), gw, gb, None
        if ctx.relu:
            g2 = torch.ops.aten.threshold_backward(g2, y, 0.0)                     # the ReLU mask
        gx = (g2 @ weight).view(ctx.x_shape) if ctx.needs_input_grad[0] else None
        gw = g2.t() @ x2 if ctx.needs_input_grad[1] else None
        gb = clip_ops.colsum(g2.contiguous()) if ctx.needs_input_grad[2] else None
        return gx, gw, gb, None


def row_linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor = None, relu: bool = False) -> torch.Tensor:
    """F.linear (``relu=True``: followed by ReLU); on CUDA fp32 tensors with a gradient to compute it runs as
    ``_RowLinear`` (same products, ReLU in the GEMM epilogue, bias gradient through the column-sum kernels -- one
    pass up to COLSUM_MAX_ROWS rows, two above; torch's multi-block reduction is kept out of captured regions, see
    models/encode_graphs.py)."""
    from ..functions import clip_ops
    if (bias is not None and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32
            and torch.is_grad_enabled() and (weight.requires_grad or x.requires_grad)
            and not torch.is_autocast_enabled() and x.numel() // max(x.shape[-1], 1) <= clip_ops.COLSUM_MAX_ROWS * clip_ops.COLSUM_CHUNK_ROWS
            and clip_ops.fused(x)):
        if relu and not FUSE_RELU_EPILOGUE:
            return torch.relu(_RowLinear.apply(x, weight, bias, False))
        return _RowLinear.apply(x, weight, bias, relu)
    y = F.linear(x, weight, bias)
    return torch.relu(y) if relu else y


def long_linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor = None, min_rows: int = None,
                activation=None) -> torch.Tensor:
    """F.linear whose weight gradient uses a split contraction when x has many rows.  ``activation`` (a module or
    function, possibly in-place) is applied to the product before it is reshaped: an in-place op on the reshaped
    VIEW of a custom Function's output makes autograd rebase the graph (CopySlices) and copy the whole gradient --
    three passes over the 22,323 x 2048 x frames FFN activation per layer (8 ms per train step, measured)."""
    rows = x.numel() // x.shape[-1]
    amp = torch.is_autocast_enabled() and x.is_cuda and bias is not None and AMP_SPLITK
    floor = AMP_MIN_ROWS if amp else (MIN_ROWS if min_rows is None else min_rows)
    if (rows >= floor and torch.is_grad_enabled() and weight.requires_grad
            and (amp or (x.dtype == weight.dtype and not torch.is_autocast_enabled()))):
        # 2-d in, 2-d out: the Function's output is then a fresh tensor (an N-d F.linear returns a view, and a
        # view made inside a custom Function may not be modified in place -- the FFN applies ReLU in place)
        fuse = (FUSE_RELU_EPILOGUE and isinstance(activation, torch.nn.ReLU) and bias is not None and x.is_cuda
                and (x.dtype == torch.float32 or amp))
        # under autocast (round 3) the same node runs with explicit casts: its bias gradient then goes through the
        # tiled column sums -- torch's multi-block reduction returned garbage inside replayed hipGraphs
        y = _SplitKLinear.apply(x.reshape(rows, x.shape[-1]), weight, bias, fuse)
        if activation is not None and not fuse:
            y = activation(y)
        return y.view(*x.shape[:-1], weight.shape[0])
    if isinstance(activation, torch.nn.ReLU):       # (never in place: an N-d linear returns a view of its product)
        return row_linear(x, weight, bias, relu=True)
    y = row_linear(x, weight, bias)
    return y if activation is None else activation(y)

