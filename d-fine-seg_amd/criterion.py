"""D-FINE criterion for the MI355X build: same loss terms, weights and key names as the
reference's DFINECriterion (src/d_fine/dfine_criterion.py:410-578), restructured so a step
never synchronises with the host:

  * all L+2 Hungarian matchings run in two HIP launches (ops.hungarian_match), the GO union
    in a third (ops.go_union); the reference does L+2 ``.cpu()`` copies + scipy calls;
  * matchings are kept dense -- ``assign[b, q]`` = matched ground-truth index or -1 -- and every
    loss is a masked reduction over fixed shapes ([B, Q, ...]), instead of gathers by
    data-dependent index lists (``pred[idx]``, ``nonzero``, ``.item()``);
  * ``num_boxes`` / ``num_boxes_go`` stay on the device and are all-reduced together as one
    2-float message (the reference issues two all-reduces and two ``.item()``).
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .configs import CriterionSpec
from .decoder import weighting_function
from .targets import TargetBatch

BATCH_MASK_HEADS = os.environ.get("DFINE_BATCH_MASK_HEADS", "1") != "0"  # measurement switch
_DUMMY_BOX = (0.5, 0.5, 1.0, 1.0)  # stands in for "no target" so IoU/GIoU stay finite before masking
_dummy_cache = {}


def _dummy_box(device):
    """Device constant, created once (never inside a graph capture: it is a host->device copy)."""
    key = str(device)
    if key not in _dummy_cache:
        _dummy_cache[key] = torch.tensor(_DUMMY_BOX, dtype=torch.float32, device=device)
    return _dummy_cache[key]


def _xyxy(b):
    cx, cy, w, h = b.unbind(-1)
    w, h = w.clamp(min=0), h.clamp(min=0)
    return cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h


def paired_iou_giou(a, b):
    """Element-wise IoU and GIoU of cxcywh boxes a, b [..., 4] (arch/utils.py:12-51, diagonal only)."""
    ax0, ay0, ax1, ay1 = _xyxy(a)
    bx0, by0, bx1, by1 = _xyxy(b)
    inter = (torch.min(ax1, bx1) - torch.max(ax0, bx0)).clamp(min=0) * (torch.min(ay1, by1) - torch.max(ay0, by0)).clamp(min=0)
    union = (ax1 - ax0) * (ay1 - ay0) + (bx1 - bx0) * (by1 - by0) - inter
    iou = inter / union
    hull = (torch.max(ax1, bx1) - torch.min(ax0, bx0)).clamp(min=0) * (torch.max(ay1, by1) - torch.min(ay0, by0)).clamp(min=0)
    return iou, iou - (hull - union) / hull


class _Matched:
    """Dense view of one matching: for every (image, query) the matched GT's box/label, or a
    dummy where unmatched."""

    def __init__(self, assign, tb: TargetBatch, num_classes):
        self.mask = assign >= 0  # [B, Q]
        idx = assign.clamp(min=0).long()
        box = tb.boxes.gather(1, idx.unsqueeze(-1).expand(-1, -1, 4))
        self.boxes = torch.where(self.mask.unsqueeze(-1), box, _dummy_box(box.device))
        self.labels = torch.where(self.mask, tb.labels.gather(1, idx), num_classes)


class DFINECriterion(nn.Module):
    def __init__(self, spec: CriterionSpec = None, num_classes=80, losses=("vfl", "boxes", "local"), label_smoothing=0.0,
                 fused=True):
        """``fused``: on the GPU, compute vfl / boxes / local of each head (and their gradients) with the
        single HIP kernel behind ``ops.criterion_head``; the dense PyTorch formulation below is the
        same math op by op and is what the kernel is tested against."""
        super().__init__()
        self.fused = fused
        self.spec = spec or CriterionSpec()
        self.num_classes = num_classes
        self.losses = tuple(losses)
        self.weight_dict = dict(self.spec.weight_dict)
        self.label_smoothing = label_smoothing  # only used by the reference's inactive focal loss

    # ------------------------------------------------------------------ loss terms (dense)
    def loss_vfl(self, out, m: _Matched, num_boxes):
        """Varifocal loss, dfine_criterion.py:92-122."""
        logits = out["pred_logits"].float()
        iou = paired_iou_giou(out["pred_boxes"].float(), m.boxes)[0].detach() * m.mask
        onehot = F.one_hot(m.labels, self.num_classes + 1)[..., :-1]
        target = iou.unsqueeze(-1) * onehot
        p = torch.sigmoid(logits).detach()
        weight = self.spec.alpha * p.pow(self.spec.gamma) * (1 - onehot) + target
        loss = F.binary_cross_entropy_with_logits(logits, target, weight=weight, reduction="sum")
        return {"loss_vfl": loss / num_boxes}

    def loss_boxes(self, out, m: _Matched, num_boxes):
        """L1 + GIoU on matched pairs, dfine_criterion.py:124-143."""
        src = out["pred_boxes"].float()
        l1 = ((src - m.boxes).abs().sum(-1) * m.mask).sum()
        giou = paired_iou_giou(src, m.boxes)[1]
        return {"loss_bbox": l1 / num_boxes, "loss_giou": ((1 - giou) * m.mask).sum() / num_boxes}

    def _fgl_targets(self, ref, tgt_boxes, fdr):
        """bbox2distance + translate_gt (arch/utils.py:267-354), element-wise over [B, Q, 4 edges]."""
        reg_max = self.spec.reg_max
        W, rs = fdr["W"], fdr["reg_scale"]
        x0, y0, x1, y1 = _xyxy(tgt_boxes)
        wq, hq = ref[..., 2] / rs + 1e-16, ref[..., 3] / rs + 1e-16
        d = torch.stack([(ref[..., 0] - x0) / wq, (ref[..., 1] - y0) / hq, (x1 - ref[..., 0]) / wq, (y1 - ref[..., 1]) / hq], -1) - 0.5 * rs
        idx = (W <= d.unsqueeze(-1)).sum(-1) - 1  # index of the closest bin on the left
        inside = (idx >= 0) & (idx < reg_max)
        li = idx.clamp(0, reg_max - 1)
        dl, dr = (d - W[li]).abs(), (W[li + 1] - d).abs()
        w_right = torch.where(inside, dl / (dl + dr), (idx >= reg_max).to(d.dtype))
        w_left = torch.where(inside, 1.0 - w_right, (idx < 0).to(d.dtype))
        left = torch.where(idx >= reg_max, reg_max - 1, li)  # reg_max-0.1 truncates to reg_max-1
        return left, w_left, w_right

    def loss_local(self, out, m: _Matched, num_boxes, cache, is_dn):
        """FGL + DDF, dfine_criterion.py:145-237 (T = 5)."""
        if "pred_corners" not in out:
            return {}
        nb, T = self.spec.reg_max + 1, self.spec.ddf_temperature
        corners = out["pred_corners"].float()
        B, Q = corners.shape[:2]
        corners = corners.view(B, Q, 4, nb)
        edge_mask = m.mask.unsqueeze(-1)
        iou = paired_iou_giou(out["pred_boxes"].float(), m.boxes)[0].detach()
        key = "fgl_dn" if is_dn else "fgl"
        if key not in cache:  # from the first head of the group, reused by the others (:157-176)
            cache[key] = self._fgl_targets(out["ref_points"].float().detach(), m.boxes, cache["fdr"])
        left, w_left, w_right = cache[key]
        logp = F.log_softmax(corners, dim=-1)
        ce = -(logp.gather(-1, left.unsqueeze(-1)).squeeze(-1) * w_left + logp.gather(-1, (left + 1).unsqueeze(-1)).squeeze(-1) * w_right)
        losses = {"loss_fgl": (ce * iou.unsqueeze(-1) * edge_mask).sum() / num_boxes}

        teacher = out.get("teacher_corners")
        if teacher is not None:
            if out.get("self_distilled"):  # a layer distilled from itself: exactly zero (:197-198)
                losses["loss_ddf"] = corners.sum() * 0
                return losses
            t_prob = F.softmax(teacher.float().detach().view(B, Q, 4, nb) / T, dim=-1)
            kl = F.kl_div(F.log_softmax(corners / T, dim=-1), t_prob, reduction="none").sum(-1)  # [B,Q,4]
            w = torch.where(m.mask, iou, out["teacher_logits"].float().detach().sigmoid().max(-1).values)
            per_edge = w.unsqueeze(-1) * (T**2) * kl
            n_pos_edges = edge_mask.sum() * 4.0
            n_neg_edges = B * Q * 4.0 - n_pos_edges
            if not is_dn:  # cached for the dn heads of the same step (:223-230)
                scale = 8.0 / B
                cache["num_pos"], cache["num_neg"] = (n_pos_edges * scale).sqrt(), (n_neg_edges * scale).sqrt()
            pos = (per_edge * edge_mask).sum() / n_pos_edges.clamp(min=1)
            neg = (per_edge * ~edge_mask).sum() / n_neg_edges.clamp(min=1)
            losses["loss_ddf"] = (pos * cache["num_pos"] + neg * cache["num_neg"]) / (cache["num_pos"] + cache["num_neg"])
        return losses

    def loss_masks(self, out, slots, tb: TargetBatch, cache):
        """Focal-BCE + dice on the matched queries only, dfine_criterion.py:272-357, dense: ``slots`` =
        (query index [B, M] int64, valid [B, M] bool, groups): target slot m of image b is ground truth
        ``m % G`` (M = groups * G) and is predicted by query ``q[b, m]``. The logits are formed here, for the
        matched rows only, from the head's mask embeddings and the shared feature map."""
        if "pred_mask_embed" not in out:
            return {}
        embed, feat = out["pred_mask_embed"], cache["mask_feat"]
        if id(embed) in cache.get("mask_losses", {}):  # every head at once: _batched_mask_stats
            return cache["mask_losses"][id(embed)]
        q, valid, groups = slots
        B, M = q.shape
        Hm, Wm = feat.shape[-2:]
        if "mask_targets" not in cache:
            cache["mask_targets"] = tb.dense_mask_targets(Hm, Wm, feat.device)  # [B, G, Hm, Wm] fp32
            cache["mask_fg"] = cache["mask_targets"].flatten(2).mean(2)  # [B, G]
        tgt, fg = cache["mask_targets"], cache["mask_fg"]
        G = tgt.shape[1]
        alpha = 0.5 + 0.25 * (1 - 2 * fg).clamp(-1, 1)  # [B, G]
        rows = embed.gather(1, q.unsqueeze(-1).expand(-1, -1, embed.shape[-1]))  # [B, M, C]
        pred = ops.mask_logits_rows(rows, feat)  # [B, M, Hm, Wm], M = groups * G
        if self.fused and pred.is_cuda and (Hm * Wm) % 8 == 0 and B * M <= 65535:
            # HIP: per-row sums of the focal terms, p*t and p in one pass (and one pass backward)
            stats = ops.mask_row_stats(pred.reshape(B, M, Hm * Wm), tgt.reshape(B, G, Hm * Wm), alpha).view(B, groups, G, 3)
            focal = stats[..., 0] / float(Hm * Wm)
            inter, psum = stats[..., 1], stats[..., 2]
            tsum = (fg * float(Hm * Wm)).unsqueeze(1)
        else:
            pred = pred.float().view(B, groups, G, Hm, Wm)
            tgt5 = tgt.unsqueeze(1)  # broadcast over the denoising groups
            a5 = alpha.view(B, 1, G, 1, 1)
            p = torch.sigmoid(pred)
            bce = F.binary_cross_entropy_with_logits(pred, tgt5.expand_as(pred), reduction="none")
            p_t = p * tgt5 + (1 - p) * (1 - tgt5)
            focal = ((a5 * tgt5 + (1 - a5) * (1 - tgt5)) * (1 - p_t) ** 2 * bce).mean((3, 4))  # [B, groups, G]
            inter, psum, tsum = (p * tgt5).sum((3, 4)), p.sum((3, 4)), tgt5.sum((3, 4))
        dice = 1.0 - (2.0 * inter + 1e-6) / (psum + tsum + 1e-6)
        w = valid.view(B, groups, G).float()
        n = w.sum().clamp(min=1)  # mean over the matched pairs, like the reference's .mean() over gathered rows
        zero_if_none = (w.sum() > 0).float()
        return {"loss_mask_bce": (focal * w).sum() / n * zero_if_none, "loss_mask_dice": (dice * w).sum() / n * zero_if_none}

    def _batched_mask_stats(self, jobs, tb: TargetBatch, cache):
        """The matched rows of EVERY head against the shared feature map in one GEMM and one statistics launch: the
        reference forms ``einsum(embed_rows, mask_feat)`` once per head (dfine_criterion.py:272-357 under every aux / dn
        output), which reads the [B, C, H/4, W/4] feature map once per head forward and twice backward and leaves autograd
        to add the heads' feature-map gradients one by one (8 heads at 640^2: 7 adds over a 420 MB map). Rows of all heads
        are concatenated along M (row m still belongs to ground truth m % G), so the per-row results are the same numbers.
        Fills cache["mask_losses"][id(embed)] = the head's two loss terms; leaves it empty when the fused kernel does not
        apply (the per-head path then runs)."""
        feat = cache["mask_feat"]
        jobs = [(out["pred_mask_embed"], pairs) for out, pairs in jobs if pairs is not None and "pred_mask_embed" in out]
        if not (self.fused and feat is not None and feat.is_cuda and len(jobs) > 1):
            return
        B = feat.shape[0]
        Hm, Wm = feat.shape[-2:]
        total = sum(int(p[0].shape[1]) for _, p in jobs)
        if (Hm * Wm) % 8 != 0 or B * total > 65535 or len({e.dtype for e, _ in jobs}) != 1:
            return
        if "mask_targets" not in cache:
            cache["mask_targets"] = tb.dense_mask_targets(Hm, Wm, feat.device)  # [B, G, Hm, Wm] fp32
            cache["mask_fg"] = cache["mask_targets"].flatten(2).mean(2)  # [B, G]
        tgt, fg = cache["mask_targets"], cache["mask_fg"]
        G = tgt.shape[1]
        alpha = 0.5 + 0.25 * (1 - 2 * fg).clamp(-1, 1)
        rows = torch.cat([e.gather(1, q.unsqueeze(-1).expand(-1, -1, e.shape[-1])) for e, (q, _, _) in jobs], 1)  # [B, sum M, C]
        if ops.mask_rows_loss_supported(rows, feat):  # GEMM + row statistics in one kernel: the logits are never written
            stats = ops.mask_rows_loss(rows, feat, tgt.reshape(B, G, Hm * Wm), alpha).view(B, total // G, G, 3)
        else:
            pred = ops.mask_logits_rows(rows, feat)
            stats = ops.mask_row_stats(pred.reshape(B, total, Hm * Wm), tgt.reshape(B, G, Hm * Wm), alpha).view(B, total // G, G, 3)
        # ... and the small tail (dice ratio, means over the matched pairs) for all heads in one set of [B, sum groups, G]
        # operations: per head it is ~15 launches forward and ~25 backward of a few hundred elements each
        focal = stats[..., 0] / float(Hm * Wm)
        inter, psum = stats[..., 1], stats[..., 2]
        tsum = (fg * float(Hm * Wm)).unsqueeze(1)
        dice = 1.0 - (2.0 * inter + 1e-6) / (psum + tsum + 1e-6)
        w = torch.cat([valid.view(B, groups, G) for _, (_, valid, groups) in jobs], 1).float()  # [B, sum groups, G]
        onehot = torch.zeros(total // G, len(jobs), dtype=torch.float32, device=feat.device)  # [sum groups, heads]; device-side
        g0 = 0                                                                                 # fills only: capturable
        for i, (_, (_, _, groups)) in enumerate(jobs):
            onehot[g0 : g0 + groups, i] = 1.0
            g0 += groups
        per_group = torch.stack([(focal * w).sum((0, 2)), (dice * w).sum((0, 2)), w.sum((0, 2))], 0)  # [3, sum groups]
        per_head = (per_group.unsqueeze(-1) * onehot.unsqueeze(0)).sum(1)  # [3, heads]; not `@`: a 3 x 54 x 9 product is not worth a library GEMM launch
        n = per_head[2].clamp(min=1)  # mean over the matched pairs, like the reference's .mean() over gathered rows
        live = (per_head[2] > 0).float()
        bce, dice_h = (per_head[0] / n * live).unbind(0), (per_head[1] / n * live).unbind(0)
        cache["mask_losses"] = {id(e): {"loss_mask_bce": bce[i], "loss_mask_dice": dice_h[i]} for i, (e, _) in enumerate(jobs)}

    # ------------------------------------------------------------------ forward
    @staticmethod
    def _host_pairs(num_gts, device):
        """(image, gt) index of every real ground truth, built on the host from the counts."""
        b = torch.cat([torch.full((n,), i, dtype=torch.int64) for i, n in enumerate(num_gts)] or [torch.zeros(0, dtype=torch.int64)])
        g = torch.cat([torch.arange(n, dtype=torch.int64) for n in num_gts] or [torch.zeros(0, dtype=torch.int64)])
        return b.to(device, non_blocking=True), g.to(device, non_blocking=True)

    def forward(self, outputs, targets=None):
        """match() -> the one collective of the criterion -> compute_losses(). A captured data-parallel step runs the
        two halves in separate graphs with the all-reduce between their replays (train.TrainStep)."""
        state = self.match(outputs, targets)
        self.reduce_normalisers(state)
        return self.compute_losses(outputs, state)

    def match(self, outputs, targets=None):
        """All matchings of the step at once (matcher.py:52-130 x (L+2)) and the local normaliser counts:
        state["norm"] = [matched queries of the GO union, ground truths] on this rank (fp32 [2])."""
        spec, C = self.spec, self.num_classes
        dev = outputs["pred_logits"].device
        tb = outputs.get("targets")
        if tb is None:
            tb = TargetBatch.from_list(targets, C, dev)
        aux, enc = outputs["aux_outputs"], outputs["enc_aux_outputs"]
        main = {k: outputs[k] for k in ("pred_logits", "pred_boxes", "pred_corners", "ref_points", "pred_mask_embed") if k in outputs}
        heads = [main] + list(aux) + [outputs["pre_outputs"]] + list(enc)
        logits = torch.stack([h["pred_logits"].detach().float() for h in heads])
        boxes = torch.stack([h["pred_boxes"].detach().float() for h in heads])
        gt_of_query, query_of_gt = ops.hungarian_match(
            logits, boxes, tb.labels, tb.boxes, tb.counts, spec.match_alpha, spec.match_gamma,
            spec.cost_class, spec.cost_bbox, spec.cost_giou)
        go = ops.go_union(gt_of_query)
        norm = torch.stack([(go >= 0).sum().float(), tb.counts.sum().float()])
        return {"tb": tb, "main": main, "gt_of_query": gt_of_query, "query_of_gt": query_of_gt, "go": go, "norm": norm}

    @staticmethod
    def reduce_normalisers(state):
        """num_boxes / num_boxes_go averaged over the ranks: ONE 2-float all-reduce, in place on state["norm"]
        (dfine_criterion.py:436-453 issues two all-reduces and two .item())."""
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(state["norm"])
            state["norm"].div_(dist.get_world_size())
        return state

    def compute_losses(self, outputs, state):
        spec, C = self.spec, self.num_classes
        dev = outputs["pred_logits"].device
        tb, main, gt_of_query, query_of_gt, go = state["tb"], state["main"], state["gt_of_query"], state["query_of_gt"], state["go"]
        aux, enc = outputs["aux_outputs"], outputs["enc_aux_outputs"]
        norm = state["norm"].clamp(min=1)
        num_boxes_go, num_boxes = norm[0], norm[1]

        fdr = outputs.get("fdr")
        if fdr is None:  # outputs not produced by this package's decoder: read the Parameters (one sync)
            fdr = {"W": weighting_function(spec.reg_max, outputs["up"], outputs["reg_scale"]).to(dev),
                   "reg_scale": float(outputs["reg_scale"].detach().abs().reshape(-1)[0])}
        with_masks = "masks" in self.losses and "mask_feat" in outputs and (tb.masks is not None or tb.mask_tgt is not None)
        cache, losses = {"fdr": fdr, "mask_feat": outputs.get("mask_feat")}, {}

        # ---- denoising heads: deterministic matching (get_cdn_matched_indices, :610-632), dense
        has_dn = "dn_outputs" in outputs
        if has_dn:
            meta = outputs["dn_meta"]
            groups = meta["dn_num_group"]
            dn_nb = num_boxes * groups if groups > 0 else torch.ones_like(num_boxes)
            dn_assign = meta["dn_assign"]
            if dn_assign is None:
                dn_assign = torch.full(outputs["dn_pre_outputs"]["pred_logits"].shape[:2], -1, dtype=torch.int64, device=dev)
            dn_pairs = None
            if with_masks and groups > 0:
                # positive slot of ground truth g in group k is denoising query k*2G + g (G = padded GT width)
                G = tb.max_gt
                q_dn = (torch.arange(groups, device=dev)[:, None] * 2 * G + torch.arange(G, device=dev)[None, :]).flatten()
                dn_pairs = (q_dn[None, :].expand(tb.valid.shape[0], -1), tb.valid.repeat(1, groups), groups)

        def own_pairs(h):
            if not with_masks:
                return None
            return query_of_gt[h].long().clamp(min=0), tb.valid, 1

        def mask_terms(out, pairs, suffix):
            if with_masks and pairs is not None:
                res = self.loss_masks(out, pairs, tb, cache)
                losses.update({k + suffix: v * self.weight_dict[k] for k, v in res.items() if k in self.weight_dict})

        if self.fused and dev.type == "cuda":
            head = self._fused_heads(tb, go, gt_of_query, num_boxes, num_boxes_go, fdr, losses)
        else:
            head = self._dense_heads(tb, go, gt_of_query, num_boxes, num_boxes_go, cache, losses)

        if with_masks:  # every head's mask logits from one GEMM (same call order as below)
            jobs = [(main, own_pairs(0))] + [(o, own_pairs(1 + i)) for i, o in enumerate(aux)]
            jobs += [(outputs["pre_outputs"], own_pairs(1 + len(aux)))] + [(o, own_pairs(2 + len(aux) + i)) for i, o in enumerate(enc)]
            if has_dn:
                jobs += [(o, dn_pairs) for o in outputs["dn_outputs"]]
                if "dn_pred_mask_embed" in outputs:
                    jobs.append(({"pred_mask_embed": outputs["dn_pred_mask_embed"]}, dn_pairs))
                jobs.append((outputs["dn_pre_outputs"], dn_pairs))
            if BATCH_MASK_HEADS:
                self._batched_mask_stats(jobs, tb, cache)

        # order of the reference's loss dict: main, aux_i, pre, enc_i, dn_i, dn_final (masks), dn_pre.
        # On the HIP path the heads are parked and computed by ONE launch when the block below ends (ops.CriterionBatch): the
        # terms they put into ``losses`` are read only by the final stack
        import contextlib

        batch = ops.CriterionBatch() if (self.fused and dev.type == "cuda" and ops.CRITERION_BATCH) else contextlib.nullcontext()
        with batch:
            self._all_heads(head, mask_terms, main, aux, enc, outputs, own_pairs, dn_pairs, has_dn, with_masks,
                            dn_assign if has_dn else None, dn_nb if has_dn else None)
        # the reference's per-term nan_to_num (dfine_criterion.py:576-578) as ONE launch over the stacked terms
        names = list(losses.keys())
        clean = torch.nan_to_num(torch.stack([losses[k] for k in names]), nan=0.0)
        return dict(zip(names, clean.unbind(0)))

    @staticmethod
    def _all_heads(head, mask_terms, main, aux, enc, outputs, own_pairs, dn_pairs, has_dn, with_masks, dn_assign, dn_nb):
        head(main, "", 0, local=True)
        mask_terms(main, own_pairs(0), "")
        for i, out in enumerate(aux):
            head(out, f"_aux_{i}", 1 + i, local=True)
            mask_terms(out, own_pairs(1 + i), f"_aux_{i}")
        head(outputs["pre_outputs"], "_pre", 1 + len(aux), local=True)
        mask_terms(outputs["pre_outputs"], own_pairs(1 + len(aux)), "_pre")
        for i, out in enumerate(enc):
            head(out, f"_enc_{i}", 2 + len(aux) + i, local=False)
            mask_terms(out, own_pairs(2 + len(aux) + i), f"_enc_{i}")
        if has_dn:
            for i, out in enumerate(outputs["dn_outputs"]):
                head(out, f"_dn_{i}", None, local=True, dn=(dn_assign, dn_nb))
                mask_terms(out, dn_pairs, f"_dn_{i}")
            if "dn_pred_mask_embed" in outputs and with_masks:
                mask_terms({"pred_mask_embed": outputs["dn_pred_mask_embed"]}, dn_pairs, "_dn_final")  # same tensor as in ``jobs``: id() keyed
            head(outputs["dn_pre_outputs"], "_dn_pre", None, local=True, dn=(dn_assign, dn_nb))
            mask_terms(outputs["dn_pre_outputs"], dn_pairs, "_dn_pre")

    # ------------------------------------------------------------------ per-head loss terms: HIP
    def _fused_heads(self, tb, go, gt_of_query, num_boxes, num_boxes_go, fdr, losses):
        """vfl + boxes + local of a head, and their gradients, in ONE kernel launch
        (ops.criterion_head); the normalisers and DDF coefficients are device scalars."""
        spec = self.spec
        dev = go.device
        key = ("loss_weights", str(dev))
        if key not in self.__dict__:  # built once (a host->device copy: never inside a graph capture)
            order = ("loss_vfl", "loss_bbox", "loss_giou", "loss_fgl", "loss_ddf")
            self.__dict__[key] = torch.tensor([float(self.weight_dict[k]) for k in order], dtype=torch.float32, device=dev)
        weights = self.__dict__[key]
        B, Q = go.shape
        n_pos = 4.0 * (go >= 0).sum().float()  # matched / unmatched edges of the GO matching
        n_neg = 4.0 * B * Q - n_pos
        w_pos, w_neg = (n_pos * (8.0 / B)).sqrt(), (n_neg * (8.0 / B)).sqrt()  # num_pos / num_neg (:223-230)
        share_pos, share_neg = w_pos / (w_pos + w_neg), w_neg / (w_pos + w_neg)
        scal_main = torch.cat([torch.stack([1.0 / num_boxes, 1.0 / num_boxes_go, share_pos / n_pos.clamp(min=1),
                                            share_neg / n_neg.clamp(min=1)]), weights])
        state = {}

        def head(out, suffix, h, local, dn=None):
            if dn is None:
                own, both, scal = gt_of_query[h], go, scal_main
            else:
                dn_assign, dn_nb = dn
                if dn_assign.shape[1] == 0:  # no denoising queries in this batch: every term is zero
                    zero = num_boxes * 0
                    names = ["loss_vfl", "loss_bbox", "loss_giou"] + (["loss_fgl", "loss_ddf"] if "pred_corners" in out else [])
                    losses.update({n + suffix: zero for n in names})
                    return
                if "scal_dn" not in state:
                    p = 4.0 * (dn_assign >= 0).sum().float()
                    n = 4.0 * dn_assign.shape[0] * dn_assign.shape[1] - p
                    state["scal_dn"] = torch.cat([torch.stack([1.0 / dn_nb, 1.0 / dn_nb, share_pos / p.clamp(min=1),
                                                               share_neg / n.clamp(min=1)]), weights])
                    state["dn_assign"] = dn_assign.to(torch.int32).contiguous()
                own = both = state["dn_assign"]
                scal = state["scal_dn"]
            corners = out.get("pred_corners") if local else None
            distil = corners is not None and out.get("teacher_corners") is not None and not out.get("self_distilled")
            terms = ops.criterion_head(
                out["pred_logits"], out["pred_boxes"], corners, out.get("ref_points"),
                out["teacher_corners"] if distil else None, out["teacher_logits"] if distil else None,
                own, both, tb.labels, tb.boxes, fdr["W"], scal, spec.alpha, spec.gamma, fdr["reg_scale"], spec.ddf_temperature)
            names = ["loss_vfl", "loss_bbox", "loss_giou"]
            if corners is not None:
                names.append("loss_fgl")
                if out.get("teacher_corners") is not None:
                    names.append("loss_ddf")  # exactly 0 for the self-distilled layer, like the reference
            losses.update({n + suffix: t for n, t in zip(names, terms)})

        return head

    # ------------------------------------------------------------------ per-head loss terms: dense PyTorch
    def _dense_heads(self, tb, go, gt_of_query, num_boxes, num_boxes_go, cache, losses):
        C = self.num_classes
        m_go = _Matched(go, tb, C)
        state = {}

        def head(out, suffix, h, local, dn=None):
            if dn is None:
                own, nb_own, other, nb_other = _Matched(gt_of_query[h], tb, C), num_boxes, m_go, num_boxes_go
            else:
                if "m_dn" not in state:
                    state["m_dn"] = _Matched(dn[0], tb, C)
                own = other = state["m_dn"]
                nb_own = nb_other = dn[1]
            res = dict(self.loss_vfl(out, own, nb_own))
            res.update(self.loss_boxes(out, other, nb_other))
            if local:
                res.update(self.loss_local(out, other, nb_other, cache, dn is not None))
            losses.update({k + suffix: v * self.weight_dict[k] for k, v in res.items() if k in self.weight_dict})

        return head
