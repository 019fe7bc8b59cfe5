"""The reference's one native operator, in mmcv's own signature.

`ms_deform_attn_forward` has the argument list of mmcv's
`ext_module.ms_deform_attn_forward` (mmcv/ops/multi_scale_deform_attn.py, called from
`MultiScaleDeformableAttnFunction.forward`; configured for Pair-Net at
configs/mask2former/pairnet.py:43-54) and runs the gfx950 kernel behind
`pn_msda_loc_f32` (csrc/msda.hip); `ms_deform_attn_backward` has the argument list of
`ext_module.ms_deform_attn_backward` and runs `pn_msda_bwd_f32`;
`MultiScaleDeformableAttnFunction` is mmcv's autograd function of the same name over the two
(what `MultiScaleDeformableAttention.forward` calls when `value.is_cuda`).  INTEGRATION.md
shows how a maintainer points mmcv's module at them.  No CPU path.
"""
import torch

from . import hip


@torch.no_grad()
def ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                           sampling_locations, attention_weights, im2col_step=64):
    """value (bs, num_keys, num_heads, embed_dims // num_heads); value_spatial_shapes
    (num_levels, 2) int64 (h, w); value_level_start_index (num_levels,) int64;
    sampling_locations (bs, num_queries, num_heads, num_levels, num_points, 2) in [0, 1],
    (x, y); attention_weights (bs, num_queries, num_heads, num_levels, num_points);
    `im2col_step` is accepted and ignored (a batching knob of the CUDA kernel).
    Returns (bs, num_queries, embed_dims)."""
    bs, n, heads, dims = value.shape
    _, nq, _, levels, points, _ = sampling_locations.shape
    if (heads, dims, points) != (8, 32, 4) or not 1 <= levels <= 4:
        raise NotImplementedError("built for 8 heads x 32 channels, 4 points, <= 4 levels "
                                  "(configs/mask2former/pairnet.py:43-54)")
    if not value.is_cuda:
        raise RuntimeError("ms_deform_attn_forward runs on an MI355X only; there is no CPU path")
    value = value.contiguous()
    out = torch.empty(bs, nq, heads * dims, device=value.device, dtype=torch.float32)
    with torch.cuda.device(value.device):
        hip.msda_loc(value, heads * dims, value_spatial_shapes.contiguous(),
                     value_level_start_index.contiguous(), sampling_locations.contiguous(),
                     attention_weights.contiguous(), out, bs, n, nq, levels)
    return out


def _check_shapes(value, sampling_locations):
    bs, n, heads, dims = value.shape
    _, nq, _, levels, points, _ = sampling_locations.shape
    if (heads, dims, points) != (8, 32, 4) or not 1 <= levels <= 4:
        raise NotImplementedError("built for 8 heads x 32 channels, 4 points, <= 4 levels "
                                  "(configs/mask2former/pairnet.py:43-54)")
    if not value.is_cuda:
        raise RuntimeError("the deformable-attention operator runs on an MI355X only; there is "
                           "no CPU path")
    return bs, n, nq, levels


@torch.no_grad()
def ms_deform_attn_backward(value, value_spatial_shapes, value_level_start_index,
                            sampling_locations, attention_weights, grad_output, grad_value,
                            grad_sampling_loc, grad_attn_weight, im2col_step=64):
    """mmcv's `ext_module.ms_deform_attn_backward`: the three gradient tensors are the caller's
    (mmcv allocates them with `torch.zeros_like`) and are filled in place -- `grad_value`
    (bs, num_keys, num_heads, dims) is ACCUMULATED into (atomics, like mmcv's col2im), so it
    must arrive zeroed; `grad_sampling_loc` and `grad_attn_weight` (the shapes of
    `sampling_locations` / `attention_weights`) are overwritten.  `im2col_step` is accepted and
    ignored.  Returns None."""
    bs, n, nq, levels = _check_shapes(value, sampling_locations)
    for t, like in ((grad_value, value), (grad_sampling_loc, sampling_locations),
                    (grad_attn_weight, attention_weights)):
        if t.shape != like.shape or t.dtype != torch.float32 or not t.is_contiguous() \
                or t.device != value.device:
            raise RuntimeError("gradient buffers: contiguous fp32 device tensors shaped like "
                               "value / sampling_locations / attention_weights")
    with torch.cuda.device(value.device):
        hip.msda_bwd(value.contiguous(), 256, value_spatial_shapes.contiguous(),
                     value_level_start_index.contiguous(), sampling_locations.contiguous(),
                     attention_weights.contiguous(), grad_output.contiguous(), grad_value,
                     grad_sampling_loc, grad_attn_weight, bs, n, nq, levels)


class MultiScaleDeformableAttnFunction(torch.autograd.Function):
    """mmcv.ops.multi_scale_deform_attn.MultiScaleDeformableAttnFunction on the gfx950 kernels:
    `apply(value, value_spatial_shapes, value_level_start_index, sampling_locations,
    attention_weights, im2col_step)` -> (bs, num_queries, embed_dims), differentiable in
    value, sampling_locations and attention_weights."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step=64):
        ctx.im2col_step = im2col_step
        out = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                     sampling_locations, attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        value, shapes, starts, loc, aw = ctx.saved_tensors
        grad_value = torch.zeros_like(value, memory_format=torch.contiguous_format)
        grad_loc = torch.zeros_like(loc, memory_format=torch.contiguous_format)
        grad_aw = torch.zeros_like(aw, memory_format=torch.contiguous_format)
        ms_deform_attn_backward(value, shapes, starts, loc, aw, grad_output.contiguous(),
                                grad_value, grad_loc, grad_aw, ctx.im2col_step)
        return grad_value, None, None, grad_loc, grad_aw, None
