"""Drop-in for the reference's compiled extension `MultiScaleDeformableAttention`
(dinounet/dinov3/eval/segmentation/models/utils/ops/src/vision.cpp:18-21), imported by name at
ms_deform_attn.py:18.  Same two entry points, argument order, output shapes, ownership (inputs borrowed, outputs
freshly allocated and zero-filled) and error behaviour (RuntimeError for non-contiguous / non-GPU tensors or a batch not
divisible by min(batch, im2col_step), ms_deform_attn_cuda.cu:33-57), backed by libdinounet_hip.so on gfx950.
fp32 and fp64 tensors (what the reference dispatches, ms_deform_attn_cuda.cu:69,139: its acceptance script ops/test.py runs gradcheck in
double) and additionally bf16 `value`.
"""
import ctypes as _C

import torch

from dinounet_amd import _lib, ops


def _f64(named):
    """all floating inputs double -> the fp64 kernels (du_msda_*_f64)"""
    return all(t.dtype == torch.float64 for _, t in named)


def _pp(t):
    return _C.c_void_p(t.data_ptr())


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
    if _f64((("value", value), ("sampling_loc", sampling_loc), ("attn_weight", attn_weight))):
        N, S, M, D = value.shape
        _, Lq, _, L, P, _ = sampling_loc.shape
        out = torch.empty((N, Lq, M * D), dtype=torch.float64, device=value.device)
        _lib.check(_lib.lib().du_msda_forward_f64(_pp(value), _pp(spatial_shapes), _pp(level_start_index), _pp(sampling_loc), _pp(attn_weight), _pp(out),
                                                  N, S, M, D, L, Lq, P, _C.c_void_p(torch.cuda.current_stream().cuda_stream)), "du_msda_forward_f64")
        return out
    return ops.msda_forward_raw(value, spatial_shapes, level_start_index, sampling_loc.float(), attn_weight.float())


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight] with the shapes/dtypes of the corresponding inputs."""
    _check((("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
            ("sampling_loc", sampling_loc), ("attn_weight", attn_weight), ("grad_output", grad_output)), im2col_step, value.shape[0])
    if _f64((("value", value), ("sampling_loc", sampling_loc), ("attn_weight", attn_weight), ("grad_output", grad_output))):
        N, S, M, D = value.shape
        _, Lq, _, L, P, _ = sampling_loc.shape
        gv, gl, ga = torch.empty_like(value), torch.empty_like(sampling_loc), torch.empty_like(attn_weight)
        _lib.check(_lib.lib().du_msda_backward_f64(_pp(value), _pp(spatial_shapes), _pp(level_start_index), _pp(sampling_loc), _pp(attn_weight),
                                                   _pp(grad_output), _pp(gv), _pp(gl), _pp(ga), N, S, M, D, L, Lq, P,
                                                   _C.c_void_p(torch.cuda.current_stream().cuda_stream)), "du_msda_backward_f64")
        return [gv, gl, ga]
    gv, gl, ga = ops.msda_backward_raw(value, spatial_shapes, level_start_index, sampling_loc.float(), attn_weight.float(),
                                       grad_output.to(value.dtype))
    return [gv.to(value.dtype), gl.to(sampling_loc.dtype), ga.to(attn_weight.dtype)]
