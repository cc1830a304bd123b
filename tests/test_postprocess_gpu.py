"""Evaluation post-processing kernels (csrc/postprocess.hip via dfine_seg_amd.postprocess.preds_postprocess) against the
reference's Trainer.preds_postprocess (tests/golden/postprocess.npz, generated from src/dl/train.py:227-319): records per
image, with / without masks, with / without letterbox.

Bar: labels and kept counts exact; scores 1e-6 (sigmoid round-off); boxes exact up to 1e-3 px (the reference mixes float32
and float64 numpy arithmetic, reproduced in the kernel); masks: integer output, >= 99.9 % of the pixels identical (the
remaining pixels sit within fp16 round-off of the 0.5 threshold: the reference resizes on the CPU with a different
summation order inside one bilinear tap)."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("keep_ratio", [False, True])
@pytest.mark.parametrize("with_masks", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32])
def test_preds_postprocess_matches_reference(hip_lib, keep_ratio, with_masks, dtype):
    from dfine_seg_amd.postprocess import preds_postprocess

    g = load_golden("postprocess")
    logits = torch.as_tensor(g["logits"]).cuda().to(dtype)
    boxes = torch.as_tensor(g["boxes"]).cuda()
    outputs = {"pred_logits": logits, "pred_boxes": boxes}
    if with_masks:
        outputs["pred_masks"] = torch.as_tensor(g["masks"]).cuda()
    proc = [int(v) for v in g["proc"]]
    inputs = torch.zeros(logits.shape[0], 3, *proc, device="cuda")
    orig_sizes = torch.as_tensor(g["orig_sizes"])
    res = preds_postprocess(inputs, outputs, orig_sizes, logits.shape[-1], keep_ratio, float(g["conf_thresh"]),
                            num_top_queries=int(g["num_top_queries"]), to_cpu=True)
    tag = f"kr{int(keep_ratio)}.m{int(with_masks)}"
    for b, r in enumerate(res):
        ref = {k[len(f"{tag}.{b}."):]: torch.as_tensor(v) for k, v in g.items() if k.startswith(f"{tag}.{b}.")}
        assert set(r.keys()) == set(ref.keys()), (sorted(r.keys()), sorted(ref.keys()))
        assert torch.equal(r["all_labels"], ref["all_labels"]) and torch.equal(r["labels"], ref["labels"])
        torch.testing.assert_close(r["all_scores"], ref["all_scores"], rtol=0, atol=1e-6)
        torch.testing.assert_close(r["scores"], ref["scores"], rtol=0, atol=1e-6)
        torch.testing.assert_close(r["all_boxes"], ref["all_boxes"].float(), rtol=0, atol=1e-3)
        torch.testing.assert_close(r["boxes"], ref["boxes"].float(), rtol=0, atol=1e-3)
        if "masks" in ref:
            assert r["masks"].dtype == torch.uint8 and r["masks"].shape == ref["masks"].shape
            same = (r["masks"] == ref["masks"]).float().mean().item()
            assert same >= 0.999, same
            assert int(r["masks"].sum()) > 0  # the fixture has foreground


def test_topk_handles_ties_and_many_classes(hip_lib):
    """Property test at the real size (Q*C = 300*80, K = 300): the kernel's selection equals a stable descending sort of
    the sigmoid scores (ties -> lower flat index first), for random logits and for heavily tied logits."""
    from dfine_seg_amd.postprocess import preds_postprocess

    g = torch.Generator(device="cuda").manual_seed(0)
    for tied in (False, True):
        logits = torch.randn(4, 300, 80, device="cuda", generator=g)
        if tied:
            logits = (logits * 2).round() / 2  # few distinct values -> thousands of ties at the threshold
        boxes = torch.rand(4, 300, 4, device="cuda", generator=g) * 0.4 + 0.3
        inputs = torch.zeros(4, 3, 640, 640, device="cuda")
        res = preds_postprocess(inputs, {"pred_logits": logits, "pred_boxes": boxes}, torch.tensor([[640, 640]] * 4), 80, False, 0.5)
        scores = torch.sigmoid(logits).flatten(1)
        order = torch.sort(scores, dim=1, descending=True, stable=True).indices[:, :300]
        for b, r in enumerate(res):
            torch.testing.assert_close(r["all_scores"], scores[b, order[b]], rtol=0, atol=1e-6)
            if not tied:
                assert torch.equal(r["all_labels"], order[b] % 80)
            else:  # among equal scores the kernel keeps the lowest flat indices, in increasing order
                assert torch.equal(r["all_labels"], order[b] % 80)
