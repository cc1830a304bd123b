"""Oracle restatements of the reference's free functions in src/d_fine/arch/utils.py.

fp32, CPU, written for clarity rather than speed. TEST INFRASTRUCTURE ONLY (see
oracle/__init__.py).
"""
import math
from typing import List, Sequence

import torch


# --------------------------------------------------------------------------------------
# deformable attention core -- reference src/d_fine/arch/utils.py:191-264 (method="default")
# --------------------------------------------------------------------------------------
def deformable_attention_core(
    value: torch.Tensor,
    value_spatial_shapes: Sequence[Sequence[int]],
    sampling_locations: torch.Tensor,
    attention_weights: torch.Tensor,
    num_points_list: List[int],
) -> torch.Tensor:
    """value [B, S, nH, c] (channel-last decoder memory), sampling_locations
    [B, Lq, nH, P, 2] in [0,1], attention_weights [B, Lq, nH, P] -> [B, Lq, nH*c].

    The reference computes, per level, ``F.grid_sample(value_l, 2*loc-1, bilinear, zeros,
    align_corners=False)`` (utils.py:217-233) and sums the samples weighted by the attention
    weights (utils.py:258-262). grid_sample's un-normalisation of ``g = 2*loc-1`` with
    align_corners=False is ``pixel = ((g+1)*size-1)/2 = loc*size-0.5``; corners outside the map
    contribute zero. That is written out here with explicit gathers.
    """
    B, S, nH, c = value.shape
    Lq = sampling_locations.shape[1]
    value = value.float()
    out = value.new_zeros(B, Lq, nH, c)
    b_idx = torch.arange(B).view(B, 1, 1, 1)
    h_idx = torch.arange(nH).view(1, 1, nH, 1)
    start, p0 = 0, 0
    for (H, W), n_pts in zip(value_spatial_shapes, num_points_list):
        loc = sampling_locations[:, :, :, p0 : p0 + n_pts].float()  # [B,Lq,nH,n,2]
        w = attention_weights[:, :, :, p0 : p0 + n_pts].float()  # [B,Lq,nH,n]
        x = loc[..., 0] * W - 0.5
        y = loc[..., 1] * H - 0.5
        x0, y0 = torch.floor(x), torch.floor(y)
        fx, fy = x - x0, y - y0
        level = value[:, start : start + H * W]  # [B, H*W, nH, c]
        for dy, wy in ((0, 1 - fy), (1, fy)):
            for dx, wx in ((0, 1 - fx), (1, fx)):
                xi, yi = x0 + dx, y0 + dy
                inside = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
                row = (yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)).long()  # [B,Lq,nH,n]
                corner = level[b_idx, row, h_idx]  # [B,Lq,nH,n,c]
                weight = (wx * wy * w * inside.to(w.dtype)).unsqueeze(-1)
                out = out + (corner * weight).sum(3)
        start += H * W
        p0 += n_pts
    return out.reshape(B, Lq, nH * c)
