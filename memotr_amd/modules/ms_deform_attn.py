"""Multi-scale deformable attention module on the gfx950 operator.

Same constructor, parameter names (``sampling_offsets``, ``attention_weights``, ``value_proj``,
``output_proj`` -- state-dict compatible), initialisation and forward contract as the reference
module (``models/ops/modules/ms_deform_attn.py:36-130``).  What differs is how the work is issued:
the offset and attention-logit projections share one GEMM over the query (their weight rows are
stacked: 256 -> 3*M*L*P outputs), and the gather/interpolate/reduce runs in the hand-written HIP
kernels behind ``MSDeformAttnFunction``.
"""
from __future__ import annotations

import math
import os
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, xavier_uniform_

from .. import MultiScaleDeformableAttention as MSDA
from ..functions import BankSlices, MSDeformAttnFunction, MSDeformAttnFusedFunction, ValueBank
from .linear import long_linear

# Fused prologue (softmax + location arithmetic + mask fill inside the HIP kernels).  On by default
# for CUDA tensors; MEMOTR_FUSED_PROLOGUE=0 (or setting this attribute) keeps the reference-shaped operator boundary.
FUSED_PROLOGUE = os.environ.get("MEMOTR_FUSED_PROLOGUE", "1") != "0"


def _is_power_of_2(n) -> bool:
    if (not isinstance(n, int)) or (n < 0):
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return (n & (n - 1) == 0) and n != 0


def _level_sizes(spatial_shapes: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """(L, 2) float (W, H) of an (L, 2) int64 (H, W) tensor, cached on the tensor object (the transformer keeps one
    ``spatial_shapes`` tensor per pyramid geometry)."""
    cached = getattr(spatial_shapes, "_msda_wh", None)
    if cached is None or cached[0] != (spatial_shapes._version, dtype):
        cached = ((spatial_shapes._version, dtype), spatial_shapes.flip(-1).to(dtype))
        try:
            spatial_shapes._msda_wh = cached
        except AttributeError:      # tensor subclasses without a __dict__
            pass
    return cached[1]


MASKED_ROWS_ATTR = "_msda_masked_rows"


def tag_masked_rows(mask: torch.Tensor) -> torch.Tensor:
    """Attach to a padding mask (N, S) the flat indices of its padded rows (one host read-back, so: callers that cache
    the mask per image geometry -- ``DeformableTransformer.encode``).  A module that finds the tag zeroes exactly those
    rows of ``value`` where the projection GEMM wrote them (and of ``grad_value`` on the way back) and calls the
    kernels WITHOUT a mask: same results as the reference's ``value.masked_fill(mask[..., None], 0)``
    (models/ops/modules/ms_deform_attn.py:107-108), but the kernels lose the dependent mask-byte loads in front of
    their gathers -- measured at the encoder shape with the 800 x 1333 frame's own mask (1344-wide padding):
    forward 73.4 -> 56.6 us, backward 174.4 -> 163.1 us per image (profiles/r04_lib_ab_mask.txt)."""
    if mask is not None and mask.is_cuda and not torch.cuda.is_current_stream_capturing():
        setattr(mask, MASKED_ROWS_ATTR, mask.reshape(-1).nonzero().squeeze(1))
    return mask


class _ZeroRows(torch.autograd.Function):
    """``x[rows] = 0`` -- in place when ``x`` is a freshly produced tensor (the 2-d product inside ``long_linear``: an
    in-place op on a VIEW of a custom Function's output would make autograd rebase the graph and copy the whole
    gradient), a copy otherwise; the gradient of those rows is zero.  ``consumer_zeroes``: the one consumer of the
    result (the fused operator, given the same rows) zeroes those rows of the gradient in the buffer it allocates --
    the backward here is then the identity.  Otherwise the incoming gradient is NOT this node's to modify (a retained
    gradient, a hook or a second consumer may hold it): the rows are zeroed in a copy."""

    @staticmethod
    def forward(ctx, x, rows, consumer_zeroes=None):
        ctx.rows = rows
        ctx.consumer_zeroes = consumer_zeroes       # a one-element list the caller sets once the consumer is known
        if x._is_view() or not x.is_contiguous():
            return x.reshape(-1, x.shape[-1]).index_fill(0, rows, 0).view(x.shape)
        x.view(-1, x.shape[-1]).index_fill_(0, rows, 0)
        ctx.mark_dirty(x)
        return x

    @staticmethod
    def backward(ctx, g):
        if ctx.consumer_zeroes is not None and ctx.consumer_zeroes[0]:
            return g, None, None
        return g.reshape(-1, g.shape[-1]).index_fill(0, ctx.rows, 0).view(g.shape), None, None


class ProjectedValue:
    """What ``project_values`` hands one module: its (N, S, M, D) slice of the shared projection, the bank that collects
    the slices' gradients, the slice's index, and the padding mask the kernels still have to apply (None when the
    padded rows were zeroed where the projection wrote them)."""
    __slots__ = ("value", "bank", "index", "mask")

    def __init__(self, value, bank, index, mask):
        self.value, self.bank, self.index, self.mask = value, bank, index, mask


BATCHED_VALUE_PROJ = os.environ.get("MEMOTR_BATCHED_VALUE_PROJ", "1") != "0"


def project_values(modules, input_flatten: torch.Tensor, input_padding_mask=None):
    """The value projections of several modules that attend to the SAME memory -- the six decoder layers of a frame,
    reference models/deformable_decoder.py:303-310 -> ms_deform_attn.py:104 -- as one GEMM with the weights stacked:
    (S x C) x (C x G C) instead of G products forward, one (S x G C) x (G C x C) product instead of G products and G - 1
    full-size additions for the memory's gradient, one weight-gradient product instead of G.  Each module reads its C
    columns of the product in place (include/msda_hip.h, msda_next_value_pixel_stride) and accumulates its share of the
    gradient into a slice of one tensor (``ValueBank``).  Returns a list of ``ProjectedValue`` (pass one as ``value=`` to
    each module's forward) or None when the calls do not qualify -- the modules then project for themselves."""
    mods = list(modules)
    if not mods or not BATCHED_VALUE_PROJ or not FUSED_PROLOGUE or len(mods) < 2:
        return None
    m0 = mods[0]
    M, L, P, C = m0.n_heads, m0.n_levels, m0.n_points, m0.d_model
    x = input_flatten
    if not (x.is_cuda and x.dim() == 3 and x.dtype == torch.float32 and not torch.is_autocast_enabled() and C // M == 32
            and MSDA.fused_supported(x.dtype, C // M, L, P)
            and all(isinstance(m, MSDeformAttn) and not m.sigmoid_attn and (m.n_heads, m.n_levels, m.n_points, m.d_model)
                    == (M, L, P, C) and m.value_proj.weight.dtype == torch.float32 for m in mods)):
        return None
    N, S, _ = x.shape
    G = len(mods)
    if (G * C * 4) % 16 != 0 or N * S * G * C >= (1 << 29):
        return None
    mask = input_padding_mask
    rows = getattr(mask, MASKED_ROWS_ATTR, None) if mask is not None else None
    zero = None
    if rows is not None:        # (as in MSDeformAttn.forward: the padded rows zeroed where the projection writes them)
        mask = None
        if rows.numel():
            zero = lambda y: _ZeroRows.apply(y, rows, None)      # noqa: E731
    w = torch.cat([m.value_proj.weight for m in mods], 0)
    b = torch.cat([m.value_proj.bias for m in mods], 0)
    value_all = long_linear(x, w, b, activation=zero)                           # (N, S, G * C)
    bank = ValueBank(G)
    slices = BankSlices.apply(value_all.view(N, S, G, M, C // M), bank)
    return [ProjectedValue(v, bank, i, mask) for i, v in enumerate(slices)]


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, sigmoid_attn=False, visualize=False):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("MSDeformAttn: a power-of-2 head dimension (32 for MeMOTR) takes the specialised "
                          "gfx950 kernels; other sizes run the generic ones.")
        self.im2col_step = 64
        self.sigmoid_attn = sigmoid_attn
        self.d_model = d_model
        self.n_levels = n_levels
        self.n_heads = n_heads
        self.n_points = n_points
        self.visualize = visualize  # accepted for config compatibility; tensor dumps are not implemented

        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self.reset_parameters()

    def reset_parameters(self):
        """Zero offset weights, bias = 8-direction star scaled by the point index (reference :72-86)."""
        constant_(self.sampling_offsets.weight.data, 0.0)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = grid / grid.abs().max(-1, keepdim=True)[0]
        grid = grid.view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        grid = grid * torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, self.n_points, 1)
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid.reshape(-1))
        constant_(self.attention_weights.weight.data, 0.0)
        constant_(self.attention_weights.bias.data, 0.0)
        xavier_uniform_(self.value_proj.weight.data)
        constant_(self.value_proj.bias.data, 0.0)
        xavier_uniform_(self.output_proj.weight.data)
        constant_(self.output_proj.bias.data, 0.0)

    def _forward_projected(self, query, reference_points, pv: "ProjectedValue", input_spatial_shapes,
                           input_level_start_index):
        """The fused branch of ``forward`` on a slice of a shared value projection (``project_values`` checked the
        conditions of that branch)."""
        M, P = self.n_heads, self.n_points
        site = self.__dict__.get("_msda_site")
        if site is None:
            site = self.__dict__["_msda_site"] = MSDA.new_call_site()
        MSDA.set_call_site(site)
        w, b = self._fused_query_projection()
        proj = long_linear(query, w, b)
        if reference_points.shape[-1] not in (2, 4):
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
                reference_points.shape[-1]))
        out = MSDeformAttnFusedFunction.apply(pv.value, input_spatial_shapes, input_level_start_index, proj.contiguous(),
                                              reference_points.contiguous(),
                                              None if pv.mask is None else pv.mask.contiguous(), M, P, None,
                                              (pv.bank, pv.index))
        return long_linear(out, self.output_proj.weight, self.output_proj.bias)

    def _fused_query_projection(self):
        """Stacked (offsets; attention-logits) weight and bias for the single query GEMM.  The module runs once per
        frame (decoder) or per encode group within a train step, always with the same parameters, so the two
        concatenations are made once per step: the cached tensors (and the autograd edge to the four parameters)
        are reused until a parameter changes or a backward pass has flowed through them."""
        so, aw = self.sampling_offsets, self.attention_weights
        pre = getattr(so.weight, "_msda_fused_qproj", None)
        if pre is not None:
            # a capture whose flat parameter argument holds the two weights (and the two biases) next to each other
            # (models/decoder_graphs.py: paired_query_projections): the stack is a view, nothing to concatenate
            return pre
        if so.weight.is_cuda and torch.cuda.is_current_stream_capturing():
            # inside a hipGraph capture the concatenation must be part of the graph (a cached tensor would freeze the
            # weights of the first capture into every replay)
            return torch.cat((so.weight, aw.weight), 0), torch.cat((so.bias, aw.bias), 0)
        key = (id(so.weight), id(aw.weight), so.weight._version, aw.weight._version, so.bias._version,
               aw.bias._version, so.weight.device, so.weight.dtype, torch.is_grad_enabled())
        cached = self.__dict__.get("_fused_qproj")
        if cached is None or cached[0] != key:
            w = torch.cat((so.weight, aw.weight), 0)
            b = torch.cat((so.bias, aw.bias), 0)
            if w.requires_grad:
                w.register_hook(self._forget_fused_query_projection)
            cached = (key, w, b)
            self.__dict__["_fused_qproj"] = cached
        return cached[1], cached[2]

    def _forget_fused_query_projection(self, grad=None):
        self.__dict__.pop("_fused_qproj", None)      # the graph behind the cached tensors ends with this backward
        return None

    # The cache is a derived tensor (and, in training, a live autograd edge): drop it whenever the parameters can
    # have changed behind the version counters (``.data`` writes by EMA / manual updates are paired with a mode
    # switch or a load in every training loop), and never let it travel with a pickle / deepcopy of the module.
    def train(self, mode: bool = True):
        self._forget_fused_query_projection()
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self._forget_fused_query_projection()
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        self._forget_fused_query_projection()
        return super()._load_from_state_dict(*args, **kwargs)

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_fused_qproj", None)
        state.pop("_msda_site", None)
        return state

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k not in ("_fused_qproj", "_msda_site"):
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None, value: "ProjectedValue" = None):
        """
        query (N, Lq, C); reference_points (N, Lq, L, 2|4) in [0,1] incl. padding; input_flatten (N, S, C);
        input_spatial_shapes (L, 2) int64 (H, W); input_level_start_index (L,); input_padding_mask (N, S) bool.
        ``value`` (optional, no reference counterpart): this module's share of ``project_values`` -- the value
        projection has then been done together with the other modules'.
        Returns (N, Lq, C).
        """
        N, Lq, _ = query.shape
        S = input_flatten.shape[1]
        M, L, P = self.n_heads, self.n_levels, self.n_points
        if value is not None:
            return self._forward_projected(query, reference_points, value, input_spatial_shapes,
                                           input_level_start_index)
        if input_flatten.is_cuda:
            # one kernel-selection record per module: the learnt offsets differ from layer to layer
            site = self.__dict__.get("_msda_site")
            if site is None:
                site = self.__dict__["_msda_site"] = MSDA.new_call_site()
            MSDA.set_call_site(site)

        # (tag_masked_rows: the padded rows of the mask are known -- zeroed where the projection writes them, and the
        #  kernels run without a mask; the conditions are those of the fused branch below)
        mask = input_padding_mask
        rows = getattr(mask, MASKED_ROWS_ATTR, None) if mask is not None else None
        if rows is not None and not (FUSED_PROLOGUE and input_flatten.is_cuda and not self.sigmoid_attn):
            rows = None
        zero = None
        fused_zeroes = [False]      # set below when the fused operator takes `rows` and zeroes its own grad_value
        if rows is not None:
            mask = None
            if rows.numel():
                zero = lambda y: _ZeroRows.apply(y, rows, fused_zeroes)      # noqa: E731
        value = long_linear(input_flatten, self.value_proj.weight, self.value_proj.bias, activation=zero)

        # one GEMM for both query projections
        n_off = M * L * P * 2
        w, b = self._fused_query_projection()
        proj = long_linear(query, w, b)
        if proj.dtype == torch.bfloat16:
            # bf16 mixed precision (no reference counterpart; policy in DESIGN.md): GEMMs and `value` in bf16,
            # sampling locations / attention weights / index arithmetic stay fp32
            proj = proj.float()
            reference_points = reference_points.float()
        if reference_points.shape[-1] not in (2, 4):
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(
                reference_points.shape[-1]))

        if (FUSED_PROLOGUE and value.is_cuda and not self.sigmoid_attn and proj.dtype == torch.float32
                and reference_points.dtype == torch.float32
                and MSDA.fused_supported(value.dtype, self.d_model // M, L, P)):
            # softmax over L*P, ref + off / (W, H) (or the box-scaled form) and the padding-mask fill run inside the
            # kernels, forward and backward: loc / attn / the masked copy of value never reach HBM
            fused_zeroes[0] = zero is not None
            out = MSDeformAttnFusedFunction.apply(value.view(N, S, M, self.d_model // M), input_spatial_shapes,
                                                  input_level_start_index, proj.contiguous(),
                                                  reference_points.contiguous(),
                                                  None if mask is None else mask.contiguous(),
                                                  M, P, rows if zero is not None else None)
            return long_linear(out, self.output_proj.weight, self.output_proj.bias)

        if mask is not None:
            value = value.masked_fill(mask[..., None], 0.0)
        value = value.view(N, S, M, self.d_model // M)
        # views (only the last dim is split), not reshape copies: the ops below write contiguous results anyway
        offsets = proj[..., :n_off].unflatten(-1, (M, L, P, 2))
        logits = proj[..., n_off:].unflatten(-1, (M, L * P))
        if self.sigmoid_attn:
            attn = logits.sigmoid()
        else:
            attn = F.softmax(logits, -1)
        attn = attn.view(N, Lq, M, L, P)

        if reference_points.shape[-1] == 2:
            wh = _level_sizes(input_spatial_shapes, offsets.dtype)  # (L, 2) as (W, H), cached per pyramid tensor
            loc = reference_points[:, :, None, :, None, :] + offsets / wh[None, None, None, :, None, :]
        else:
            if P & (P - 1) == 0:
                # offsets / P * wh * 0.5 with P a power of two: scaling by powers of two commutes with rounding,
                # so folding 0.5 / P into the (small) reference tensor gives the same bits with two fewer kernels
                loc = reference_points[:, :, None, :, None, :2] \
                    + offsets * (reference_points[:, :, None, :, None, 2:] * (0.5 / P))
            else:
                loc = reference_points[:, :, None, :, None, :2] \
                    + offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5

        out = MSDeformAttnFunction.apply(value.contiguous(), input_spatial_shapes, input_level_start_index,
                                         loc.contiguous(), attn.contiguous(), self.im2col_step)
        return long_linear(out, self.output_proj.weight, self.output_proj.bias)
