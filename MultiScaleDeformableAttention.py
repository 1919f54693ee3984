"""Drop-in for the reference's compiled extension `MultiScaleDeformableAttention`
(dinounet/dinov3/eval/segmentation/models/utils/ops/src/vision.cpp:18-21), imported by name at
ms_deform_attn.py:18.  Same two entry points, argument order, output shapes, ownership (inputs borrowed, outputs
freshly allocated and zero-filled) and error behaviour (RuntimeError for non-contiguous / non-GPU tensors or a batch not
divisible by min(batch, im2col_step), ms_deform_attn_cuda.cu:33-57), backed by libdinounet_hip.so on gfx950.
fp32 tensors (the reference dispatches fp32/fp64 only, ms_deform_attn_cuda.cu:69) and additionally bf16 `value`.
"""
import torch

from dinounet_amd import ops


def _check(named, im2col_step, batch):
    for n, t in named:
        if not t.is_contiguous():
            raise RuntimeError(f"{n} tensor has to be contiguous")
        if not t.is_cuda:
            raise RuntimeError(f"{n} must be a CUDA tensor")
    step = min(batch, im2col_step)
    if batch % step != 0:
        raise RuntimeError(f"batch({batch}) must divide im2col_step({step})")


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """value (N,S,M,D), spatial_shapes (L,2) int64, level_start_index (L) int64, sampling_loc (N,Lq,M,L,P,2),
    attn_weight (N,Lq,M,L,P) -> (N, Lq, M*D)."""
    _check((("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
            ("sampling_loc", sampling_loc), ("attn_weight", attn_weight)), im2col_step, value.shape[0])
    return ops.msda_forward_raw(value, spatial_shapes, level_start_index, sampling_loc.float(), attn_weight.float())


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight] with the shapes/dtypes of the corresponding inputs."""
    _check((("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
            ("sampling_loc", sampling_loc), ("attn_weight", attn_weight), ("grad_output", grad_output)), im2col_step, value.shape[0])
    gv, gl, ga = ops.msda_backward_raw(value, spatial_shapes, level_start_index, sampling_loc.float(), attn_weight.float(),
                                       grad_output.to(value.dtype))
    return [gv.to(value.dtype), gl.to(sampling_loc.dtype), ga.to(attn_weight.dtype)]
