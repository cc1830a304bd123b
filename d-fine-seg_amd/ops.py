"""Host-side operators over the C ABI: autograd Functions whose forward and backward are
HIP kernels from libdfine_hip.so. Names and argument meaning follow the reference's own
operator seams so the parity tests read like calls into the reference.
"""
from typing import List, Sequence

import torch

from . import _lib


class _MSDeformAttnCore(torch.autograd.Function):
    """Gather behind decoder cross-attention; see include/dfine_hip.h (dfine_msda_fwd/bwd)."""

    @staticmethod
    def forward(ctx, value, loc, attn, level_hw, num_points):
        _lib.require_gpu(value, loc, attn)
        lib = _lib.load()
        B, S, nH, c = value.shape
        Lq = loc.shape[1]
        P = sum(num_points)
        if loc.shape != (B, Lq, nH, P, 2) or attn.shape != (B, Lq, nH, P):
            raise ValueError(
                f"sampling_locations {tuple(loc.shape)} / attention_weights {tuple(attn.shape)} "
                f"do not match value {tuple(value.shape)} with {P} points per head"
            )
        value = value.contiguous()
        loc = loc.detach().float().contiguous()
        attn = attn.detach().float().contiguous()
        out = torch.empty(B, Lq, nH * c, dtype=value.dtype, device=value.device)
        hw = _lib.i32_array(v for pair in level_hw for v in pair)
        npts = _lib.i32_array(num_points)
        rc = lib.dfine_msda_fwd(
            _lib.ptr(value), _lib.dtype_code(value), _lib.ptr(loc), _lib.ptr(attn), _lib.ptr(out),
            B, S, Lq, nH, c, len(num_points), hw, npts, _lib.stream_ptr(),
        )
        _lib.check(rc, "dfine_msda_fwd")
        ctx.save_for_backward(value, loc, attn)
        ctx.geom = (tuple(map(tuple, level_hw)), tuple(num_points))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        value, loc, attn = ctx.saved_tensors
        level_hw, num_points = ctx.geom
        lib = _lib.load()
        B, S, nH, c = value.shape
        Lq = loc.shape[1]
        grad_out = grad_out.to(value.dtype).contiguous()
        grad_value = torch.zeros(B, S, nH, c, dtype=torch.float32, device=value.device)
        grad_loc = torch.empty_like(loc)
        grad_attn = torch.empty_like(attn)
        hw = _lib.i32_array(v for pair in level_hw for v in pair)
        npts = _lib.i32_array(num_points)
        rc = lib.dfine_msda_bwd(
            _lib.ptr(value), _lib.dtype_code(value), _lib.ptr(loc), _lib.ptr(attn),
            _lib.ptr(grad_out), _lib.ptr(grad_value), _lib.ptr(grad_loc), _lib.ptr(grad_attn),
            B, S, Lq, nH, c, len(num_points), hw, npts, _lib.stream_ptr(),
        )
        _lib.check(rc, "dfine_msda_bwd")
        return grad_value.to(value.dtype), grad_loc, grad_attn, None, None


def deformable_attention_core(
    value: torch.Tensor,
    value_spatial_shapes: Sequence[Sequence[int]],
    sampling_locations: torch.Tensor,
    attention_weights: torch.Tensor,
    num_points_list: List[int],
) -> torch.Tensor:
    """MI355X replacement for ``deformable_attention_core_func_v2(..., method="default")``
    (reference src/d_fine/arch/utils.py:191-264).

    Same arguments as the reference except ``value``: the decoder memory viewed channel-last
    ``[B, sum(H_l*W_l), n_head, c]`` (float32 or bfloat16) instead of the reference's per-level
    tuple of ``(B, n_head, c, H_l*W_l)`` (use :func:`value_from_reference_layout` to convert).
    Returns ``[B, Lq, n_head*c]`` in ``value``'s dtype; fp32 accumulation inside.
    """
    if value.dtype not in (torch.float32, torch.bfloat16):
        value = value.float()
    return _MSDeformAttnCore.apply(
        value, sampling_locations, attention_weights,
        [tuple(int(v) for v in hw) for hw in value_spatial_shapes], [int(n) for n in num_points_list],
    )


def value_from_reference_layout(value_levels) -> torch.Tensor:
    """Reference ``value_op`` output (tuple of ``(B, n_head, c, H_l*W_l)``,
    dfine_decoder.py:416-426) -> channel-last ``[B, S, n_head, c]``."""
    return torch.cat(list(value_levels), dim=-1).permute(0, 3, 1, 2).contiguous()
