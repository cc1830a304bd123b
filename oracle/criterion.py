"""Hungarian matcher + D-FINE criterion, fp32 CPU restatement. TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py).

Reference: src/d_fine/matcher.py:52-130 and src/d_fine/dfine_criterion.py:92-659.
The criterion is a plain function of (outputs, targets): the reference keeps per-call caches
on the module (fgl targets, num_pos/num_neg, dfine_criterion.py:61-66,394-397); here they are
a local dict created per call.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import configs as C
from . import ops
from .lsap import linear_sum_assignment


# --------------------------------------------------------------------------------------
# matcher
# --------------------------------------------------------------------------------------
def match_cost(pred_logits, pred_boxes, tgt_ids, tgt_boxes):
    """matcher.py:76-112: cost [B*Q, sum(n_gt)] = 5*L1 + 2*focal-class + 2*(-GIoU), NaN -> 1."""
    prob = torch.sigmoid(pred_logits.flatten(0, 1).float())[:, tgt_ids]
    boxes = pred_boxes.flatten(0, 1).float()
    a, g = C.MATCH_ALPHA, C.MATCH_GAMMA
    neg = (1 - a) * prob**g * (-(1 - prob + 1e-8).log())
    pos = a * (1 - prob) ** g * (-(prob + 1e-8).log())
    cost_class = pos - neg
    cost_bbox = torch.cdist(boxes, tgt_boxes, p=1)
    cost_giou = -ops.pairwise_giou(ops.cxcywh_to_xyxy(boxes), ops.cxcywh_to_xyxy(tgt_boxes))
    w = C.MATCH_WEIGHTS
    cost = w["cost_bbox"] * cost_bbox + w["cost_class"] * cost_class + w["cost_giou"] * cost_giou
    return torch.nan_to_num(cost, nan=1.0)


@torch.no_grad()
def hungarian_match(pred_logits, pred_boxes, targets):
    """HungarianMatcher.forward, matcher.py:52-130 -> list over images of (query_idx, gt_idx) int64."""
    B, Q = pred_logits.shape[:2]
    tgt_ids = torch.cat([t["labels"] for t in targets]).long()
    tgt_boxes = torch.cat([t["boxes"] for t in targets]).float()
    cost = match_cost(pred_logits, pred_boxes, tgt_ids, tgt_boxes).view(B, Q, -1)
    sizes = [len(t["boxes"]) for t in targets]
    out = []
    for b, c in enumerate(cost.split(sizes, -1)):
        r, cidx = linear_sum_assignment(c[b].numpy())
        out.append((torch.as_tensor(r, dtype=torch.int64), torch.as_tensor(cidx, dtype=torch.int64)))
    return out


def go_indices(indices, indices_aux_list):
    """_get_go_indices, dfine_criterion.py:371-392: union of the matchings of all heads; a query
    matched to different GTs keeps the GT it was matched to most often. Ties between equally
    frequent GTs follow the reference's (count-descending argsort, first come) order only as far
    as torch's sort is deterministic; fixtures avoid relying on it."""
    results = []
    merged = [(i.clone(), j.clone()) for i, j in indices]
    for aux in indices_aux_list:
        merged = [(torch.cat([a[0], b[0]]), torch.cat([a[1], b[1]])) for a, b in zip(merged, aux)]
    for q_idx, g_idx in merged:
        pairs = torch.stack([q_idx, g_idx], 1)
        uniq, counts = torch.unique(pairs, return_counts=True, dim=0)
        order = torch.argsort(counts, descending=True)
        chosen = {}
        for q, g in uniq[order].tolist():
            chosen.setdefault(q, g)
        results.append((torch.tensor(list(chosen.keys()), dtype=torch.int64), torch.tensor(list(chosen.values()), dtype=torch.int64)))
    return results


def cdn_matched_indices(dn_meta, targets):
    """get_cdn_matched_indices, dfine_criterion.py:610-632."""
    out = []
    for i, t in enumerate(targets):
        n = len(t["labels"])
        if n > 0:
            out.append((dn_meta["dn_positive_idx"][i], torch.arange(n, dtype=torch.int64).tile(dn_meta["dn_num_group"])))
        else:
            out.append((torch.zeros(0, dtype=torch.int64), torch.zeros(0, dtype=torch.int64)))
    return out


# --------------------------------------------------------------------------------------
# loss terms
# --------------------------------------------------------------------------------------
def _src_idx(indices):
    b = torch.cat([torch.full_like(s, i) for i, (s, _) in enumerate(indices)])
    return b, torch.cat([s for s, _ in indices])


def _matched_gt(targets, indices, key):
    return torch.cat([t[key][j] for t, (_, j) in zip(targets, indices)], 0)


def _diag_iou(a_cxcywh, b_cxcywh):
    return torch.diag(ops.pairwise_iou(ops.cxcywh_to_xyxy(a_cxcywh), ops.cxcywh_to_xyxy(b_cxcywh))[0])


def loss_vfl(out, targets, indices, num_boxes, num_classes):
    """loss_labels_vfl, dfine_criterion.py:92-122."""
    idx = _src_idx(indices)
    logits = out["pred_logits"].float()
    ious = _diag_iou(out["pred_boxes"].float()[idx], _matched_gt(targets, indices, "boxes")).detach()
    cls = torch.full(logits.shape[:2], num_classes, dtype=torch.int64)
    cls[idx] = _matched_gt(targets, indices, "labels").long()
    onehot = F.one_hot(cls, num_classes + 1)[..., :-1]
    score = torch.zeros(logits.shape[:2], dtype=logits.dtype)
    score[idx] = ious
    target_score = score.unsqueeze(-1) * onehot
    p = torch.sigmoid(logits).detach()
    weight = C.VFL_ALPHA * p.pow(C.VFL_GAMMA) * (1 - onehot) + target_score
    loss = F.binary_cross_entropy_with_logits(logits, target_score, weight=weight, reduction="none")
    return {"loss_vfl": loss.mean(1).sum() * logits.shape[1] / num_boxes}


def loss_boxes(out, targets, indices, num_boxes):
    """loss_boxes, dfine_criterion.py:124-143."""
    idx = _src_idx(indices)
    src = out["pred_boxes"].float()[idx]
    tgt = _matched_gt(targets, indices, "boxes")
    giou = torch.diag(ops.pairwise_giou(ops.cxcywh_to_xyxy(src), ops.cxcywh_to_xyxy(tgt)))
    return {"loss_bbox": (src - tgt).abs().sum() / num_boxes, "loss_giou": (1 - giou).sum() / num_boxes}


def loss_local(out, targets, indices, num_boxes, cache, T=5):
    """loss_local (FGL + DDF), dfine_criterion.py:145-237; unimodal_distribution_focal_loss :638-659."""
    losses = {}
    if "pred_corners" not in out:
        return losses
    nb = C.REG_MAX + 1
    idx = _src_idx(indices)
    tgt_boxes = _matched_gt(targets, indices, "boxes")
    is_dn = "is_dn" in out
    pred = out["pred_corners"].float()[idx].reshape(-1, nb)
    ref = out["ref_points"].float()[idx].detach()
    key = "fgl_dn" if is_dn else "fgl"
    if key not in cache:  # computed once from the FIRST call (main / first dn layer), reused by all layers
        cache[key] = ops.bbox2distance(ref, ops.cxcywh_to_xyxy(tgt_boxes), C.REG_MAX, out["reg_scale"], out["up"])
    t_corners, w_right, w_left = cache[key]
    ious = _diag_iou(out["pred_boxes"].float()[idx], tgt_boxes)
    w_t = ious.unsqueeze(-1).repeat(1, 1, 4).reshape(-1).detach()
    left = t_corners.long()
    fgl = F.cross_entropy(pred, left, reduction="none") * w_left + F.cross_entropy(pred, left + 1, reduction="none") * w_right
    losses["loss_fgl"] = (fgl * w_t).sum() / num_boxes

    if "teacher_corners" in out:
        pc = out["pred_corners"].float().reshape(-1, nb)
        tc = out["teacher_corners"].float().reshape(-1, nb)
        if torch.equal(pc, tc):
            losses["loss_ddf"] = pc.sum() * 0
        else:
            w_local = out["teacher_logits"].float().sigmoid().max(-1)[0]
            mask = torch.zeros_like(w_local, dtype=torch.bool)
            mask[idx] = True
            mask = mask.unsqueeze(-1).repeat(1, 1, 4).reshape(-1)
            w_local[idx] = ious.reshape_as(w_local[idx]).to(w_local.dtype)
            w_local = w_local.unsqueeze(-1).repeat(1, 1, 4).reshape(-1).detach()
            kl = F.kl_div(F.log_softmax(pc / T, dim=1), F.softmax(tc.detach() / T, dim=1), reduction="none").sum(-1)
            per_edge = w_local * (T**2) * kl
            if not is_dn:
                scale = 8 / out["pred_boxes"].shape[0]
                cache["num_pos"], cache["num_neg"] = (mask.sum() * scale) ** 0.5, ((~mask).sum() * scale) ** 0.5
            l_pos = per_edge[mask].mean() if mask.any() else 0
            l_neg = per_edge[~mask].mean() if (~mask).any() else 0
            losses["loss_ddf"] = (l_pos * cache["num_pos"] + l_neg * cache["num_neg"]) / (cache["num_pos"] + cache["num_neg"])
    return losses


def loss_masks(out, targets, indices, num_boxes):
    """loss_masks, dfine_criterion.py:314-357 with _prepare_target_masks :239-270,
    _focal_loss_mask :272-297, _dice_loss :299-312."""
    if "pred_masks" not in out:
        return {}
    pm = out["pred_masks"].float()
    Hm, Wm = pm.shape[-2:]
    b_idx, q_idx = _src_idx(indices)
    if b_idx.numel() == 0:
        z = pm.sum() * 0
        return {"loss_mask_bce": z, "loss_mask_dice": z}
    pred = pm[b_idx, q_idx]
    tgts = []
    for t, (_, J) in zip(targets, indices):
        m = t.get("masks")
        if m is None or m.numel() == 0 or m.dim() != 3 or J.numel() == 0:
            continue
        tgts.append(F.interpolate(m[J].unsqueeze(1).float(), size=(Hm, Wm), mode="nearest").squeeze(1).clamp_(0, 1))
    if not tgts:
        z = pred.sum() * 0
        return {"loss_mask_bce": z, "loss_mask_dice": z}
    tgt = torch.cat(tgts, 0)
    assert pred.shape[0] == tgt.shape[0]
    fg = tgt.flatten(1).mean(1, keepdim=True).unsqueeze(-1)
    alpha = 0.5 + 0.25 * (1 - 2 * fg).clamp(-1, 1)
    p = torch.sigmoid(pred)
    bce = F.binary_cross_entropy_with_logits(pred, tgt, reduction="none")
    p_t = p * tgt + (1 - p) * (1 - tgt)
    alpha_t = alpha * tgt + (1 - alpha) * (1 - tgt)
    focal = (alpha_t * (1 - p_t) ** 2.0 * bce).mean(dim=(1, 2)).mean()
    pf, tf = p.flatten(1), tgt.flatten(1)
    dice = (1.0 - (2.0 * (pf * tf).sum(1) + 1e-6) / (pf.sum(1) + tf.sum(1) + 1e-6)).mean()
    return {"loss_mask_bce": focal, "loss_mask_dice": dice}


# --------------------------------------------------------------------------------------
# DFINECriterion.forward -- dfine_criterion.py:410-578
# --------------------------------------------------------------------------------------
def criterion(outputs, targets, num_classes=80, with_masks=False, world_size=1, return_indices=False):
    """Returns the weighted loss dict (same keys as the reference). world_size>1 is not
    restated (the num_boxes all-reduce is identity at world_size 1)."""
    losses_on = ["vfl", "boxes", "local"] + (["masks"] if with_masks else [])
    W = C.LOSS_WEIGHTS
    cache = {}
    match = lambda o: hungarian_match(o["pred_logits"], o["pred_boxes"], targets)

    indices = match(outputs)
    cached, cached_enc, aux_list = [], [], []
    for o in outputs["aux_outputs"] + [outputs["pre_outputs"]]:
        m = match(o)
        cached.append(m), aux_list.append(m)
    for o in outputs["enc_aux_outputs"]:
        m = match(o)
        cached_enc.append(m), aux_list.append(m)
    idx_go = go_indices(indices, aux_list)
    num_boxes_go = max(float(sum(len(x[0]) for x in idx_go)) / world_size, 1.0)
    num_boxes = max(float(sum(len(t["labels"]) for t in targets)) / world_size, 1.0)

    def terms(o, own_idx, suffix, go_for=("boxes", "local"), idx_override=None, nb_override=None):
        res = {}
        for name in losses_on:
            use_go = name in go_for and idx_override is None
            idx = idx_override if idx_override is not None else (idx_go if use_go else own_idx)
            nb = nb_override if nb_override is not None else (num_boxes_go if use_go else num_boxes)
            if name == "vfl":
                d = loss_vfl(o, targets, idx, nb, num_classes)
            elif name == "boxes":
                d = loss_boxes(o, targets, idx, nb)
            elif name == "local":
                d = loss_local(o, targets, idx, nb, cache)
            else:
                d = loss_masks(o, targets, idx, nb)
            res.update({k + suffix: v * W[k] for k, v in d.items() if k in W})
        return res

    losses = terms(outputs, indices, "")
    for i, o in enumerate(outputs["aux_outputs"]):
        o = dict(o, up=outputs["up"], reg_scale=outputs["reg_scale"])
        losses.update(terms(o, cached[i], f"_aux_{i}"))
    losses.update(terms(outputs["pre_outputs"], cached[-1], "_pre"))
    for i, o in enumerate(outputs["enc_aux_outputs"]):
        losses.update(terms(o, cached_enc[i], f"_enc_{i}", go_for=("boxes",)))

    if "dn_outputs" in outputs:
        idx_dn = cdn_matched_indices(outputs["dn_meta"], targets)
        dn_nb = num_boxes * outputs["dn_meta"]["dn_num_group"]
        dn_nb = dn_nb if dn_nb > 0 else 1
        for i, o in enumerate(outputs["dn_outputs"]):
            o = dict(o, is_dn=True, up=outputs["up"], reg_scale=outputs["reg_scale"])
            losses.update(terms(o, None, f"_dn_{i}", idx_override=idx_dn, nb_override=dn_nb))
        if "dn_pred_masks" in outputs and with_masks:
            d = loss_masks({"pred_masks": outputs["dn_pred_masks"]}, targets, idx_dn, dn_nb)
            losses.update({k + "_dn_final": v * W[k] for k, v in d.items() if k in W})
        if "dn_pre_outputs" in outputs:
            losses.update(terms(outputs["dn_pre_outputs"], None, "_dn_pre", idx_override=idx_dn, nb_override=dn_nb))

    losses = {k: torch.nan_to_num(v, nan=0.0) if torch.is_tensor(v) else torch.tensor(float(v)) for k, v in losses.items()}
    if return_indices:
        return losses, dict(indices=indices, aux=cached, enc=cached_enc, go=idx_go)
    return losses
