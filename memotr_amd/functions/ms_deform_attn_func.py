"""Autograd binding of the gfx950 operator.

Mirror of ``MSDeformAttnFunction`` in the reference
(``models/ops/functions/ms_deform_attn_func.py:24-41``): six positional arguments,
gradients for ``value``, ``sampling_locations`` and ``attention_weights`` only, no double
backward.  There is deliberately no ``ms_deform_attn_core_pytorch`` here: the pure-PyTorch
statement of the operator lives in ``oracle/`` as test infrastructure.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import MultiScaleDeformableAttention as MSDA


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
                im2col_step):
        ctx.im2col_step = im2col_step
        ctx.site = MSDA.get_call_site()       # the backward runs on autograd's thread: carry the module's tag over
        # keep the python-side copy of the pyramid with the graph node: the backward must not have to read
        # spatial_shapes back from the device if the saved tensor comes back as a fresh python object
        ctx.shapes_host = getattr(value_spatial_shapes, "_msda_host", None)
        output = MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                             sampling_locations, attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights = ctx.saved_tensors
        if ctx.shapes_host is not None and getattr(value_spatial_shapes, "_msda_host", None) is None:
            value_spatial_shapes._msda_host = (ctx.shapes_host[0], value_spatial_shapes._version)
        MSDA.set_call_site(ctx.site)
        grad_value, grad_sampling_loc, grad_attn_weight = MSDA.ms_deform_attn_backward(
            value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
            grad_output.contiguous(), ctx.im2col_step)
        MSDA.set_call_site(0)
        return grad_value, None, None, grad_sampling_loc, grad_attn_weight, None


class MSDeformAttnFusedFunction(Function):
    """The operator with the module's prologue folded in (no reference counterpart):
    ``apply(value, spatial_shapes, level_start_index, proj, reference_points, padding_mask, n_heads, n_points[,
    zero_rows])`` where ``proj`` is the raw output of the two query projections, ``[offsets (M,L,P,2) | logits (M,L,P)]`` per
    query.  Softmax over the L*P logits, the location arithmetic of ``models/ops/modules/ms_deform_attn.py:113-122``
    and the padding-mask fill of ``value`` (:107-108) happen inside the HIP kernels, forward and backward:
    sampling locations and attention weights are never written to memory.  Gradients: value, proj and -- when it
    requires one -- reference_points.  ``zero_rows`` (optional, int64 indices into the N*S rows of ``value``): rows the
    caller has zeroed instead of passing a padding mask; their gradient is zeroed HERE, in the buffer this backward
    has just allocated (nobody else holds it yet -- the caller's own hook could not know that)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, proj, reference_points, padding_mask,
                n_heads, n_points, zero_rows=None, bank=None):
        ctx.n_heads, ctx.n_points = int(n_heads), int(n_points)
        ctx.zero_rows = zero_rows
        ctx.bank = bank          # (ValueBank, index): ``value`` is slice ``index`` of a wider projection (below)
        ctx.site = MSDA.get_call_site()
        ctx.shapes_host = getattr(value_spatial_shapes, "_msda_host", None)
        output = MSDA.ms_deform_attn_fused_forward(value, value_spatial_shapes, value_level_start_index, proj,
                                                   reference_points, padding_mask, ctx.n_heads, ctx.n_points)
        # (the output too: sum_j a_j dL/da_j of the softmax Jacobian is <grad_output_row, output_row>, which lets the
        #  backward finish in one kernel -- include/msda_hip.h, msda_fused_backward_out_*; the consumer's Linear keeps the
        #  tensor alive for its weight gradient anyway)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, proj, reference_points,
                              padding_mask, output)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, level_start, proj, reference_points, padding_mask, output = ctx.saved_tensors
        if ctx.shapes_host is not None and getattr(shapes, "_msda_host", None) is None:
            shapes._msda_host = (ctx.shapes_host[0], shapes._version)
        MSDA.set_call_site(ctx.site)
        gv_out = None
        if ctx.bank is not None and MSDA.value_pixel_stride(value):
            gv_out = ctx.bank[0].grad_slice(value, ctx.bank[1])
        grad_value, grad_proj, grad_ref = MSDA.ms_deform_attn_fused_backward(
            value, shapes, level_start, proj, reference_points, padding_mask, grad_output.contiguous(), ctx.n_heads,
            ctx.n_points, need_ref_grad=ctx.needs_input_grad[4], fwd_output=output, grad_value_out=gv_out)
        MSDA.set_call_site(0)
        if ctx.zero_rows is not None and ctx.zero_rows.numel():
            grad_value.view(-1, grad_value.shape[-2] * grad_value.shape[-1]).index_fill_(0, ctx.zero_rows, 0)
        return grad_value, None, None, grad_proj, grad_ref, None, None, None, None, None


class ValueBank:
    """The gradient side of ONE projection shared by G attention modules (``BankSlices``): the first backward among
    them makes the zeroed float32 (N, S, G, M, D) tensor, every module's operator accumulates into its own slice, and
    ``BankSlices.backward`` hands the whole tensor on -- no per-module gradient tensors, no additions."""

    def __init__(self, groups: int):
        self.groups = int(groups)
        self.grad = None

    def grad_slice(self, value, index: int):
        N, S, M, D = value.shape
        if self.grad is None:
            # (fill_, not torch.zeros: under capture torch.zeros of more than a few thousand elements becomes a MEMSET
            #  node, tools/zeros_probe.py -- the captured regions hold none, models/decoder_graphs.py)
            self.grad = torch.empty((N, S, self.groups, M, D), dtype=torch.float32, device=value.device).fill_(0.0)
        return self.grad[:, :, index]


class BankSlices(Function):
    """``value_all`` (N, S, G, M, D) -> its G slices (N, S, M, D) as views (rows of a wider projection, read in place by
    the kernels: include/msda_hip.h, msda_next_value_pixel_stride).  Backward: when every incoming gradient is the
    matching slice of the bank's tensor, that tensor IS the gradient; otherwise the slices are stacked."""

    @staticmethod
    def forward(ctx, value_all, bank):
        ctx.bank = bank
        ctx.dtype = value_all.dtype
        return tuple(value_all.unbind(2))

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        bank = ctx.bank
        whole, bank.grad = bank.grad, None
        if whole is not None and len(grads) == whole.shape[2] and all(
                g is not None and g.dtype == whole.dtype and g.shape == whole[:, :, i].shape
                and g.data_ptr() == whole[:, :, i].data_ptr() and g.stride() == whole[:, :, i].stride()
                for i, g in enumerate(grads)):
            return whole.to(ctx.dtype), None
        like = next((g for g in grads if g is not None), None)
        if like is None:
            return None, None
        return torch.stack([torch.zeros_like(like) if g is None else g.to(like.dtype) for g in grads], dim=2).to(ctx.dtype), None
