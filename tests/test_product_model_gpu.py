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
        assert _rel(_sample(res["pred_masks"], 4096)[0], g["train.pred_masks.sample"]) < 1e-3

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
        assert abs(got_n - ref) <= 1e-2 * ref + 1e-4 * scale, (n, got_n, ref)


@pytest.mark.parametrize("fname", ["model_n320", "model_m640_seg"])
def test_eval_forward_matches_reference(hip_lib, fname):
    g, model, _, x, _, _, seg = _setup(fname)
    model.eval()
    with torch.no_grad():
        ev = model(x)
    assert _rel(ev["pred_logits"], g["eval.pred_logits"]) < 2e-4
    assert _rel(ev["pred_boxes"], g["eval.pred_boxes"]) < 2e-4
    if seg:
        assert _rel(_sample(ev["pred_masks"], 4096)[0], g["eval.pred_masks.sample"]) < 1e-3


@pytest.mark.parametrize("fname", ["model_m640", "model_m640_seg"])
def test_bf16_autocast_forward_and_loss(hip_lib, fname):
    """The precision the step is trained and benchmarked in. Stage tolerances (relative L2):
    boxes 2e-2, logits 3e-2 (bf16 has 8 mantissa bits; ~60 layers deep), total loss 2e-2."""
    g, model, crit, x, targets, inject, seg = _setup(fname)
    model.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        res = model(x, targets=targets, inject=inject)
    assert _rel(res["pred_boxes"], g["train.pred_boxes"]) < 2e-2
    assert _rel(res["pred_logits"], g["train.pred_logits"]) < 3e-2
    assert _rel(res["enc_aux_outputs"][0]["pred_boxes"], g["train.enc0.pred_boxes"]) < 2e-2
    losses = crit(res, targets)
    total = float(sum(losses.values()))
    ref_total = float(torch.as_tensor(g["loss.values"]).sum())
    assert abs(total - ref_total) <= 2e-2 * abs(ref_total), (total, ref_total)
    sum(losses.values()).backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
