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


# --------------------------------------------------------------------------------------
# boxes -- reference src/d_fine/arch/utils.py:12-72
# --------------------------------------------------------------------------------------
def cxcywh_to_xyxy(b):
    """utils.py:59-67 (width/height clamped at 0)."""
    cx, cy, w, h = b.unbind(-1)
    w, h = w.clamp(min=0.0), h.clamp(min=0.0)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def xyxy_to_cxcywh(b):
    """utils.py:70-73."""
    x0, y0, x1, y1 = b.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], dim=-1)


def pairwise_iou(a, b):
    """utils.py:12-25: [N,4] x [M,4] xyxy -> (iou [N,M], union [N,M])."""
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = area_a[:, None] + area_b[None] - inter
    return inter / union, union


def pairwise_giou(a, b):
    """utils.py:28-51 (the degenerate-box asserts at :39-42 are not restated)."""
    iou, union = pairwise_iou(a, b)
    lt = torch.min(a[:, None, :2], b[None, :, :2])
    rb = torch.max(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    hull = wh[..., 0] * wh[..., 1]
    return iou - (hull - union) / hull


def inverse_sigmoid(x, eps=1e-5):
    """utils.py:54-56."""
    x = x.clip(0.0, 1.0)
    return torch.log(x.clip(min=eps) / (1 - x).clip(min=eps))


# --------------------------------------------------------------------------------------
# FDR: W(n), distance <-> box -- reference src/d_fine/arch/utils.py:119-188, 267-354
# --------------------------------------------------------------------------------------
def weighting_function(reg_max, up, reg_scale):
    """utils.py:145-188 (training branch): the reg_max+1 non-uniform bin positions W(n).
    `up` is a float32 tensor [1]; `reg_scale` a tensor [1] (int64 in the shipped configs) or a
    number. Evaluated in fp32 tensor arithmetic like the reference so the bins agree bit for bit."""
    up0 = torch.as_tensor(up, dtype=torch.float32).reshape(-1)[0].abs()
    rs = torch.as_tensor(reg_scale).reshape(-1)[0].abs()
    ub1 = up0 * rs
    ub2 = ub1 * 2
    step = (ub1 + 1) ** (2 / (reg_max - 2))
    left = [-(step**i) + 1 for i in range(reg_max // 2 - 1, 0, -1)]
    right = [step**i - 1 for i in range(1, reg_max // 2)]
    return torch.stack([-ub2] + left + [torch.zeros_like(ub1)] + right + [ub2]).to(torch.float32)


def integral(pred_corners, project):
    """Integral.forward, dfine_decoder.py:291-295: softmax over the 33 bins, dot with W(n)."""
    shape = pred_corners.shape
    prob = torch.softmax(pred_corners.reshape(-1, project.numel()), dim=1)
    return (prob @ project.to(prob.dtype)).reshape(*shape[:-1], 4)


def distance2bbox(points, distance, reg_scale):
    """utils.py:119-142: edge distances -> cxcywh box around `points` (cxcywh)."""
    rs = abs(float(torch.as_tensor(reg_scale).reshape(-1)[0]))
    x1 = points[..., 0] - (0.5 * rs + distance[..., 0]) * (points[..., 2] / rs)
    y1 = points[..., 1] - (0.5 * rs + distance[..., 1]) * (points[..., 3] / rs)
    x2 = points[..., 0] + (0.5 * rs + distance[..., 2]) * (points[..., 2] / rs)
    y2 = points[..., 1] + (0.5 * rs + distance[..., 3]) * (points[..., 3] / rs)
    return xyxy_to_cxcywh(torch.stack([x1, y1, x2, y2], -1))


def translate_gt(gt, reg_max, reg_scale, up):
    """utils.py:267-326: continuous edge distance -> (left bin index, w_right, w_left)."""
    gt = gt.reshape(-1)
    W = weighting_function(reg_max, up, reg_scale).to(gt.dtype)
    left_idx = (W[None, :] - gt[:, None] <= 0).sum(1) - 1
    idx = left_idx.float()
    w_right = torch.zeros_like(idx)
    w_left = torch.zeros_like(idx)
    ok = (idx >= 0) & (idx < reg_max)
    li = idx[ok].long()
    dl = (gt[ok] - W[li]).abs()
    dr = (W[li + 1] - gt[ok]).abs()
    w_right[ok] = dl / (dl + dr)
    w_left[ok] = 1.0 - w_right[ok]
    neg = idx < 0
    w_right[neg], w_left[neg], idx[neg] = 0.0, 1.0, 0.0
    pos = idx >= reg_max
    w_right[pos], w_left[pos], idx[pos] = 1.0, 0.0, reg_max - 0.1
    return idx, w_right, w_left


def bbox2distance(points, bbox, reg_max, reg_scale, up, eps=0.1):
    """utils.py:329-354: (ref cxcywh, gt xyxy) -> FGL targets."""
    rs = abs(float(torch.as_tensor(reg_scale).reshape(-1)[0]))
    l = (points[:, 0] - bbox[:, 0]) / (points[:, 2] / rs + 1e-16) - 0.5 * rs
    t = (points[:, 1] - bbox[:, 1]) / (points[:, 3] / rs + 1e-16) - 0.5 * rs
    r = (bbox[:, 2] - points[:, 0]) / (points[:, 2] / rs + 1e-16) - 0.5 * rs
    b = (bbox[:, 3] - points[:, 1]) / (points[:, 3] / rs + 1e-16) - 0.5 * rs
    d, w_right, w_left = translate_gt(torch.stack([l, t, r, b], -1), reg_max, rs, up)
    d = d.clamp(min=0, max=reg_max - eps)
    return d.reshape(-1).detach(), w_right.detach(), w_left.detach()


# --------------------------------------------------------------------------------------
# contrastive denoising group -- reference src/d_fine/arch/utils.py:357-467
# --------------------------------------------------------------------------------------
def cdn_noise(num_gts: Sequence[int], num_classes: int, num_denoising: int = 100):
    """The four random draws of the CDN group from the GLOBAL torch CPU generator, in the
    reference's order and with its call forms (utils.py:410-419: rand_like(int32 classes ->
    float), randint_like(bool mask -> int32), randint_like(float boxes, 0, 2), rand_like(boxes)),
    so that after ``torch.manual_seed(s)`` they reproduce the reference's noise bit for bit.
    Returns None when the batch has no GT (utils.py:373-376)."""
    max_gt = max(num_gts)
    if max_gt == 0:
        return None
    G = max(1, num_denoising // max_gt)
    B, N = len(num_gts), 2 * max_gt * G
    flip_u = torch.rand_like(torch.empty(B, N, dtype=torch.int32), dtype=torch.float)
    new_label = torch.randint_like(torch.empty(B, N, dtype=torch.bool), 0, num_classes, dtype=torch.int32)
    sign01 = torch.randint_like(torch.empty(B, N, 4), 0, 2)
    part_u = torch.rand_like(torch.empty(B, N, 4))
    return dict(flip_u=flip_u, new_label=new_label, sign01=sign01, part_u=part_u)


def cdn_group(targets, num_classes, num_queries, class_embed_weight, noise, num_denoising=100,
              label_noise_ratio=0.5, box_noise_scale=1.0):
    """get_contrastive_denoising_training_group with the noise injected (utils.py:357-467).
    Returns (dn_content [B,Nd,hidden], dn_bbox_unact [B,Nd,4], attn_mask [Nd+Q,Nd+Q] bool, dn_meta)."""
    num_gts = [len(t["labels"]) for t in targets]
    max_gt = max(num_gts)
    if max_gt == 0:
        return None, None, None, dict(dn_positive_idx=None, dn_num_group=0, dn_num_split=[0, num_queries])
    G = max(1, num_denoising // max_gt)
    B = len(targets)
    cls = torch.full((B, max_gt), num_classes, dtype=torch.int32)
    box = torch.zeros(B, max_gt, 4)
    pad = torch.zeros(B, max_gt, dtype=torch.bool)
    for i, t in enumerate(targets):
        n = num_gts[i]
        if n:
            cls[i, :n] = t["labels"].to(torch.int32)
            box[i, :n] = t["boxes"]
            pad[i, :n] = True
    cls, box, pad = cls.tile(1, 2 * G), box.tile(1, 2 * G, 1), pad.tile(1, 2 * G)
    neg = torch.zeros(B, 2 * max_gt, 1)
    neg[:, max_gt:] = 1
    neg = neg.tile(1, G, 1)
    pos_mask = (1 - neg).squeeze(-1) * pad
    pos_idx = torch.nonzero(pos_mask)[:, 1].split([n * G for n in num_gts])
    Nd = 2 * max_gt * G
    if label_noise_ratio > 0:
        flip = noise["flip_u"] < (label_noise_ratio * 0.5)
        cls = torch.where(flip & pad, noise["new_label"], cls)
    if box_noise_scale > 0:
        xyxy = cxcywh_to_xyxy(box)
        diff = (box[..., 2:] * 0.5).tile(1, 1, 2) * box_noise_scale
        sign = noise["sign01"] * 2.0 - 1.0
        part = (noise["part_u"] + 1.0) * neg + noise["part_u"] * (1 - neg)
        xyxy = (xyxy + sign * part * diff).clip(0.0, 1.0)
        box = xyxy_to_cxcywh(xyxy)
        box = torch.where(box < 0, -box, box)
        box_unact = inverse_sigmoid(box)
    content = torch.nn.functional.embedding(cls.long(), class_embed_weight, padding_idx=num_classes)
    T = Nd + num_queries
    mask = torch.zeros(T, T, dtype=torch.bool)
    mask[Nd:, :Nd] = True  # matching queries never see the denoising part
    for g in range(G):  # each group sees only itself (utils.py:442-455)
        lo, hi = 2 * max_gt * g, 2 * max_gt * (g + 1)
        mask[lo:hi, hi:Nd] = True
        mask[lo:hi, :lo] = True
    meta = dict(dn_positive_idx=pos_idx, dn_num_group=G, dn_num_split=[Nd, num_queries])
    return content, box_unact, mask, meta
