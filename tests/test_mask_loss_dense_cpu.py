"""The criterion's dense, matched-rows-only mask loss (logits formed from mask embeddings after matching)
against the reference's formulation on the same numbers: full `pred_masks` for every query, gather of the
matched (image, query, gt) triples, focal-BCE + dice, mean over pairs (dfine_criterion.py:272-357). CPU:
this is the PyTorch branch of `loss_masks`; the HIP branch is checked against it in test_maskloss_gpu.py."""
import torch
import torch.nn.functional as F

import dfine_seg_amd  # noqa: F401
from dfine_seg_amd.criterion import DFINECriterion
from dfine_seg_amd.dfine import build_loss
from dfine_seg_amd.targets import TargetBatch


def _reference_style(pred_masks, tgt_masks, b, q, g):
    pred = pred_masks[b, q].float()
    tgt = tgt_masks[b, g]
    fg = tgt.flatten(1).mean(1).view(-1, 1, 1)
    alpha = 0.5 + 0.25 * (1 - 2 * fg).clamp(-1, 1)
    p = torch.sigmoid(pred)
    bce = F.binary_cross_entropy_with_logits(pred, tgt, reduction="none")
    p_t = p * tgt + (1 - p) * (1 - tgt)
    focal = ((alpha * tgt + (1 - alpha) * (1 - tgt)) * (1 - p_t) ** 2 * bce).mean((1, 2)).mean()
    pf, tf = p.flatten(1), tgt.flatten(1)
    dice = (1.0 - (2.0 * (pf * tf).sum(1) + 1e-6) / (pf.sum(1) + tf.sum(1) + 1e-6)).mean()
    return focal, dice


def test_dense_mask_loss_equals_gathered_formulation():
    torch.manual_seed(0)
    B, Q, C, H, W = 3, 10, 16, 12, 12
    num_gts = [2, 0, 3]
    G = max(num_gts)
    crit = build_loss("n", 80, 0.0, True)
    assert isinstance(crit, DFINECriterion)
    embed = torch.randn(B, Q, C, requires_grad=True)
    feat = torch.randn(B, C, H, W, requires_grad=True)
    masks = [(torch.rand(n, H, W) > 0.6).to(torch.uint8) for n in num_gts]
    labels = torch.zeros(B, G, dtype=torch.int64)
    valid = torch.tensor([[g < n for g in range(G)] for n in num_gts])
    tb = TargetBatch(labels, torch.zeros(B, G, 4), valid, num_gts, torch.tensor(num_gts, dtype=torch.int32), masks)
    # own head: ground truth g of image b is predicted by a distinct query
    query_of_gt = torch.tensor([[4, 7, 0], [0, 0, 0], [1, 9, 3]])
    out = {"pred_mask_embed": embed}
    res = crit.loss_masks(out, (query_of_gt, valid, 1), tb, {"mask_feat": feat})
    pred_masks = torch.einsum("bqc,bchw->bqhw", embed, feat)
    tgt = tb.mask_targets(H, W, "cpu")[0]
    b = torch.tensor([0, 0, 2, 2, 2])
    g = torch.tensor([0, 1, 0, 1, 2])
    q = query_of_gt[b, g]
    focal, dice = _reference_style(pred_masks, tgt, b, q, g)
    torch.testing.assert_close(res["loss_mask_bce"], focal, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(res["loss_mask_dice"], dice, rtol=1e-5, atol=1e-6)
    # gradients flow to the embeddings of matched queries only
    (res["loss_mask_bce"] + res["loss_mask_dice"]).backward()
    touched = embed.grad.abs().sum(-1) > 0
    want = torch.zeros(B, Q, dtype=torch.bool)
    want[b, q] = True
    assert torch.equal(touched, want)
    assert feat.grad[1].abs().sum() == 0  # the image without ground truth contributes nothing

    # denoising heads: `groups` copies of every ground truth, slot k*G + g <- query k*2G + g
    groups = 2
    embed_dn = torch.randn(B, groups * 2 * G, C)
    q_dn = (torch.arange(groups)[:, None] * 2 * G + torch.arange(G)[None, :]).flatten()
    res_dn = crit.loss_masks({"pred_mask_embed": embed_dn}, (q_dn[None].expand(B, -1), valid.repeat(1, groups), groups), tb,
                             {"mask_feat": feat.detach()})
    pm = torch.einsum("bqc,bchw->bqhw", embed_dn, feat.detach())
    bb, gg = b.repeat(groups), g.repeat(groups)
    qq = (torch.arange(groups)[:, None] * 2 * G + g[None, :]).flatten()
    focal_dn, dice_dn = _reference_style(pm, tgt, bb, qq, gg)
    torch.testing.assert_close(res_dn["loss_mask_bce"], focal_dn, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(res_dn["loss_mask_dice"], dice_dn, rtol=1e-5, atol=1e-6)


def test_no_ground_truth_gives_zero_mask_loss():
    crit = build_loss("n", 80, 0.0, True)
    B, Q, C, H, W = 2, 5, 8, 8, 8
    tb = TargetBatch(torch.zeros(B, 1, dtype=torch.int64), torch.zeros(B, 1, 4), torch.zeros(B, 1, dtype=torch.bool), [0, 0],
                     torch.zeros(B, dtype=torch.int32), None, torch.zeros(B, 1, H, W))
    res = crit.loss_masks({"pred_mask_embed": torch.randn(B, Q, C)}, (torch.zeros(B, 1, dtype=torch.int64), tb.valid, 1), tb,
                          {"mask_feat": torch.randn(B, C, H, W)})
    assert float(res["loss_mask_bce"]) == 0.0 and float(res["loss_mask_dice"]) == 0.0
