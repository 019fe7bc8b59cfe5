"""The reference's one native operator, in mmcv's own signature.

`ms_deform_attn_forward` has the argument list of mmcv's
`ext_module.ms_deform_attn_forward` (mmcv/ops/multi_scale_deform_attn.py, called from
`MultiScaleDeformableAttnFunction.forward`; configured for Pair-Net at
configs/mask2former/pairnet.py:43-54) and runs the gfx950 kernel behind
`pn_msda_loc_f32` (csrc/msda.hip).  INTEGRATION.md shows how a maintainer points mmcv's
module at it.  Inference only (no backward); no CPU path.
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
