"""The product model + criterion on the MI355X against outputs of the reference
(tests/golden/model_*.npz): same name-hashed weights, same synthetic batch, the reference's
query selection (top-k indices) and denoising noise injected so per-index comparison is
meaningful (SURVEY.md 7 "Hard parts").

fp32 run: tolerance = kernel/op-order noise of an fp32 GPU pipeline vs fp32 CPU.
bf16 run (autocast, as trained): per-tensor relative L2 <= the bounds written below."""
import pytest
import torch

from conftest import load_golden
from oracle import ops as oracle_ops
from oracle import weights
from oracle.gen_golden import MODEL_CASES, SEED_CDN, SEED_TARGETS, SEED_WEIGHTS_IMG, _sample

pytestmark = pytest.mark.gpu


def _setup(fname):
    from dfine_seg_amd.dfine import build_loss, build_model

    size, img, n_gts, seg = MODEL_CASES[fname]
    g = load_golden(fname)
    model = build_model(size, 80, seg, "cpu", img_size=[img, img])
    model.load_state_dict(weights.fill_state_dict(model.state_dict()), strict=True)
    model = model.cuda()
    x = weights.synthetic_images(len(n_gts), img, img, SEED_WEIGHTS_IMG).cuda()
    targets = weights.synthetic_targets(n_gts, 80, SEED_TARGETS, with_masks=seg, height=img, width=img)
    torch.manual_seed(SEED_CDN)
    noise = oracle_ops.cdn_noise(n_gts, 80)
    inject = {"topk_ind": torch.as_tensor(g["topk_ind"]).cuda()}
    if noise is not None:
        inject["cdn_noise"] = {k: v.cuda() for k, v in noise.items()}
    return g, model, build_loss(size, 80, 0.0, seg), x, targets, inject, seg


def _rel(a, b):
    a, b = a.detach().float().cpu(), torch.as_tensor(b).float()
    return float((a - b).norm() / b.norm().clamp(min=1e-12))


@pytest.mark.parametrize("fname", ["model_n320", "model_n320_empty", "model_s640", "model_m640", "model_m640_seg", "model_x640"])
def test_fp32_forward_loss_grads_match_reference(hip_lib, fname):
    g, model, crit, x, targets, inject, seg = _setup(fname)
    model.train()
    res = model(x, targets=targets, inject=inject)
    for k in ("pred_logits", "pred_boxes", "pred_corners", "ref_points"):
        assert _rel(res[k], g[f"train.{k}"]) < 2e-4, k
    heads = {f"aux{i}": d for i, d in enumerate(res["aux_outputs"])}
    heads.update(pre=res["pre_outputs"], enc0=res["enc_aux_outputs"][0])
    heads.update({f"dn{i}": d for i, d in enumerate(res["dn_outputs"])})
    heads["dn_pre"] = res["dn_pre_outputs"]
    for name, d in heads.items():
        assert _rel(d["pred_boxes"], g[f"train.{name}.pred_boxes"]) < 2e-4, name
        assert _rel(_sample(d["pred_logits"], 2048)[0], g[f"train.{name}.pred_logits.sample"]) < 2e-4, name
    if seg:
        # training heads carry mask embeddings + the shared feature map; the reference's tensor on demand:
        from dfine_seg_amd.decoder import mask_logits

        pred_masks = mask_logits(res["pred_mask_embed"], res["mask_feat"])
        assert _rel(_sample(pred_masks, 4096)[0], g["train.pred_masks.sample"]) < 1e-3

    losses = crit(res, targets)
    names = [str(n) for n in g["loss.names"]]
    assert list(losses.keys()) == names
    got = torch.stack([losses[n].detach().float().reshape(()) for n in names]).cpu()
    torch.testing.assert_close(got, torch.as_tensor(g["loss.values"]), rtol=2e-3, atol=2e-4)

    sum(losses.values()).backward()
    ref_norms = {str(n): float(v) for n, v in zip(g["grad.names"], g["grad.norms"])}
    scale = max(ref_norms.values())
    params = dict(model.named_parameters())
    for n, ref in ref_norms.items():
        grad = params[n].grad
        if ref < 0:
            assert grad is None or float(grad.abs().max()) == 0.0, n
            continue
        got_n = float(grad.norm()) if grad is not None else 0.0
        # A LAB scale/bias (or any bias) feeding a conv+BatchNorm has a mathematically ~zero
        # gradient (the BN removes scale and shift): what both sides report is cancellation noise
        # of sums over 10^5..10^6 activations, compared on the global gradient scale only.
        noise = 1e-3 if ".lab." in n else 3e-4
        assert abs(got_n - ref) <= 1e-2 * ref + noise * scale, (n, got_n, ref)


@pytest.mark.parametrize("fname", ["model_n320", "model_m640_seg"])
def test_eval_forward_matches_reference(hip_lib, fname):
    g, model, _, x, _, _, seg = _setup(fname)
    model.eval()
    with torch.no_grad():
        ev = model(x)
    # Eval runs its own top-k: queries whose scores tie to ~1e-6 may swap places between two fp32
    # pipelines, so rows are compared in a canonical order (sorted by box centre) per image.
    def canon(boxes, logits):
        boxes, logits = torch.as_tensor(boxes).float().cpu(), torch.as_tensor(logits).float().cpu()
        key = (boxes[..., 0] * 1e3).round() * 1e4 + (boxes[..., 1] * 1e3).round() + boxes[..., 2]
        order = key.argsort(dim=1)
        pick = lambda t: t.gather(1, order.unsqueeze(-1).expand_as(t))
        return pick(boxes), pick(logits)

    got_b, got_l = canon(ev["pred_boxes"].detach(), ev["pred_logits"].detach())
    ref_b, ref_l = canon(g["eval.pred_boxes"], g["eval.pred_logits"])
    assert _rel(got_l, ref_l) < 5e-4
    assert _rel(got_b, ref_b) < 5e-4
    if seg:
        assert _rel(_sample(ev["pred_masks"], 4096)[0], g["eval.pred_masks.sample"]) < 1e-3


@pytest.mark.parametrize("fname", ["model_m640", "model_m640_seg"])
def test_bf16_autocast_forward_and_loss(hip_lib, fname):
    """The precision the step is trained and benchmarked in (bf16 autocast; the reference's AMP
    is fp16/bf16 autocast too, train.py:545-551).

    With random (name-hashed) weights a bf16 pipeline is far from its fp32 twin: BatchNorm over
    2 images re-normalises rounding noise layer after layer. The tolerance is therefore calibrated
    on the reference itself -- model_m640.npz holds the reference's own bf16-autocast results
    ("bf16ref.*") next to its fp32 ones: encoder maps 27-41 % relative L2 apart, encoder class
    logits 3.9 %, total loss 0.35 %. In that regime the growth of rounding noise through ~60
    BatchNorm layers is chaotic (measured on the MI355X: 0.9 % relative error entering stem2a's
    BN, 16 % leaving it), so for the encoder maps the test only asks that the product's bf16 run
    stays correlated with fp32 (relative L2 < 1; measured 0.79-0.95 vs the reference's 0.27-0.41)
    and is within 2x of the reference's distance on the encoder class logits. The tensors that
    matter for training are bounded tightly: final-layer boxes 2e-2 / logits 8e-2 relative L2
    (query selection and denoising noise injected), total loss within 2 % of the fp32
    reference. Kernel-level bf16 accuracy is pinned separately (tests/test_msda.py)."""
    g, model, crit, x, targets, inject, seg = _setup(fname)
    cal = load_golden("model_m640")
    model.train()
    grabbed = {}
    hook = model.encoder.register_forward_hook(lambda m, i, o: grabbed.__setitem__("outs", o[0]))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        res = model(x, targets=targets, inject=inject)
    hook.remove()
    for lvl, f in enumerate(grabbed["outs"]):
        ref_err = _rel(torch.as_tensor(cal[f"bf16ref.encoder.outs.{lvl}.sample"]), cal[f"encoder.outs.{lvl}.sample"])
        assert _rel(_sample(f)[0], g[f"encoder.outs.{lvl}.sample"]) < max(1.0, 1.5 * ref_err), lvl
    ref_err = _rel(torch.as_tensor(cal["bf16ref.enc0.pred_logits.sample"]), cal["train.enc0.pred_logits.sample"])
    enc_logits = _sample(res["enc_aux_outputs"][0]["pred_logits"], 2048)[0]
    assert _rel(enc_logits, g["train.enc0.pred_logits.sample"]) < 2 * ref_err
    assert _rel(res["pred_boxes"], g["train.pred_boxes"]) < 2e-2
    assert _rel(res["pred_logits"], g["train.pred_logits"]) < 8e-2
    assert _rel(res["enc_aux_outputs"][0]["pred_boxes"], g["train.enc0.pred_boxes"]) < 2e-2
    losses = crit(res, targets)
    total = float(sum(losses.values()))
    ref_total = float(torch.as_tensor(g["loss.values"]).sum())
    assert abs(total - ref_total) <= 2e-2 * abs(ref_total), (total, ref_total)
    sum(losses.values()).backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
