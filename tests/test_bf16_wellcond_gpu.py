"""The BENCHMARKED bf16 path (TrainStep amp=True: implicit-GEMM conv / Linear, HIP attention, grouped weight gradients
accumulating straight into the flat fp32 buffer, packed-bf16 MSDA value-gradient sink) pinned to the REFERENCE on a
well-conditioned fixture: tests/golden/model_m640*_wellcond.npz = the reference model (fp32, CPU) in train() mode with every
BatchNorm2d in .eval(), on oracle/weights.py's "wellcond" weights (non-negative convolutions, calibrated unit-scale norms:
nothing cancels, so rounding errors add instead of multiplying; oracle/gen_golden.py:gen_wellcond).

Why not the name-hashed default weights: they make every conv a difference of large positive terms, and ANY bf16 pipeline
drifts from fp32 by a few percent per layer -- measured with frozen statistics too (this repo's bf16 path: 5 / 12 / 26 % at
the three backbone maps, 33-81 % at the encoder's; the reference's own CPU autocast: 27-41 %, model_m640.npz bf16ref.*), so
the train-mode-BN tests in test_product_model_gpu.py can only bound that regime loosely. On the well-conditioned weights the
same kernels must agree with the reference to percent level END TO END: stage outputs, the loss dict and per-module
gradients. Query selection and the denoising noise are the reference's (injected), as everywhere else."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import ops as oracle_ops
from oracle import weights
from oracle.gen_golden import WELLCOND_CASES, SEED_CDN, SEED_TARGETS, SEED_WEIGHTS_IMG, _sample, grad_sample_index

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = torch.as_tensor(a).detach().float().cpu(), torch.as_tensor(b).float()
    return float((a - b).norm() / b.norm().clamp(min=1e-12))


def _run(fname, amp=True, msda_f32=False):
    from dfine_seg_amd import ops
    from dfine_seg_amd.targets import TargetBatch
    from dfine_seg_amd.train import TrainStep

    size, img, n_gts, seg = WELLCOND_CASES[fname]
    g = load_golden(fname)
    t = TrainStep(size, 80, seg, (img, img), "cuda", amp=amp, use_graph=False)
    sd = weights.fill_state_dict(t.model.state_dict(), "wellcond")
    off = 0
    for name, ch in zip(g["bn.names"], g["bn.channels"]):  # the calibrated norm scales of the fixture (running_mean = 0)
        sd[f"{name}.running_var"] = torch.as_tensor(g["bn.running_var"][off : off + ch])
        off += int(ch)
    t.model.load_state_dict(sd, strict=True)
    t.flat.refresh_bf16_shadow()
    for m in t.model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()
    grabbed = {}
    hooks = [t.model.backbone.register_forward_hook(lambda m, i, o: grabbed.__setitem__("backbone", o)),
             t.model.encoder.register_forward_hook(lambda m, i, o: grabbed.__setitem__("encoder", o))]
    x = weights.synthetic_images(len(n_gts), img, img, SEED_WEIGHTS_IMG).cuda().contiguous(memory_format=torch.channels_last)
    targets = weights.synthetic_targets(n_gts, 80, SEED_TARGETS, with_masks=seg, height=img, width=img)
    torch.manual_seed(SEED_CDN)
    noise = oracle_ops.cdn_noise(n_gts, 80)
    inject = {"topk_ind": torch.as_tensor(g["topk_ind"]).cuda(), "cdn_noise": {k: v.cuda() for k, v in noise.items()}}
    tb = TargetBatch.from_list(targets, 80, t.device)
    if tb.masks is not None:
        s = t.model.decoder.spec.strides[0] // 2
        tb.dense_mask_targets(img // s, img // s, t.device)
    old, old_mode = ops.MSDA_GRAD_BF16, ops.MSDA_VALUE_GRAD
    ops.MSDA_GRAD_BF16 = not msda_f32
    if not msda_f32:
        ops.MSDA_VALUE_GRAD = "atomic"  # the packed-bf16 map exists in the scatter form only
    try:
        t._forward_backward(t._model_input(x), tb, inject)
    finally:
        ops.MSDA_GRAD_BF16, ops.MSDA_VALUE_GRAD = old, old_mode
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    grads = {n: p.grad.detach().float().cpu() for n, p in t.model.named_parameters() if p.requires_grad and p.grad is not None}
    return g, t, grabbed, grads


def _grad_metrics(g, grads):
    """Per top-level module: cosine between the sampled product and reference gradients (all tensors of the module
    concatenated) and the norm-weighted relative L2 distance; plus the worst per-tensor relative L2 among the tensors that
    carry at least 1 % of their module's gradient norm."""
    names, counts = list(g["grad.names"]), g["grad.counts"]
    offs = np.concatenate([[0], np.cumsum(counts)])
    per_mod = {}
    for i, n in enumerate(names):
        ref = torch.as_tensor(g["grad.samples"][offs[i] : offs[i + 1]]).float()
        got = grads[n].flatten()[grad_sample_index(grads[n].numel())]
        per_mod.setdefault(n.split(".")[0], []).append((n, got, ref, float(g["grad.norms"][i]), float(grads[n].norm())))
    out = {}
    for mod, rows in per_mod.items():
        a, b = torch.cat([r[1] for r in rows]), torch.cat([r[2] for r in rows])
        mod_norm = float(np.sqrt(sum(r[3] ** 2 for r in rows)))
        worst = max(((float((r[1] - r[2]).norm() / r[2].norm().clamp(min=1e-20)), r[0]) for r in rows if r[3] >= 0.01 * mod_norm),
                    default=(0.0, ""))
        norm_err = float(np.sqrt(sum((r[4] - r[3]) ** 2 for r in rows)) / max(mod_norm, 1e-20))
        out[mod] = dict(cos=float(torch.dot(a, b) / (a.norm() * b.norm())), rel=float((a - b).norm() / b.norm()), worst=worst,
                        norm_err=norm_err)
    return out


@pytest.mark.parametrize("fname,sink", [("model_m640_wellcond", "default"), ("model_m640_seg_wellcond", "default"),
                                        ("model_m640_wellcond", "f32"), ("model_m640_wellcond", "bf16")])
def test_bf16_train_step_matches_reference_on_well_conditioned_weights(hip_lib, fname, sink):
    """``sink``: how the deformable attention's value gradient is accumulated -- the default, float atomics ("f32") or packed
    bf16 atomics ("bf16"; measured here, not asserted: in this fixture every query box sits at one image corner, so a few
    memory tokens receive thousands of adds each and a bf16 running sum swamps -- backbone cosine -0.5)."""
    from dfine_seg_amd import ops

    g, t, grabbed, grads = _run(fname, msda_f32={"default": not ops.MSDA_GRAD_BF16, "f32": True, "bf16": False}[sink])
    seg = bool(g["seg"])
    report = {}
    # stage outputs (bf16 against the reference's fp32): relative L2 <= 3e-2
    for lvl, f in enumerate(grabbed["backbone"]):
        report[f"backbone.{lvl}"] = _rel(_sample(f, 4096)[0], g[f"backbone.{lvl}.sample"])
    for lvl, f in enumerate(grabbed["encoder"][0]):
        report[f"encoder.{lvl}"] = _rel(_sample(f, 4096)[0], g[f"encoder.outs.{lvl}.sample"])
    # loss dict: every term the reference reports, the total within 1e-2
    got = {k: float(v) for k, v in t.last_losses.items()}
    names = [str(n) for n in g["loss.names"]]
    assert list(got) == names
    ref_total, total = float(g["loss.values"].sum()), sum(got.values())
    report["loss.total"] = abs(total - ref_total) / abs(ref_total)
    report["loss.worst_term"] = max(abs(got[n] - float(v)) / max(abs(float(v)), 0.05) for n, v in zip(names, g["loss.values"]))
    gm = _grad_metrics(g, grads)
    print(f"\n[{fname}] " + " ".join(f"{k}={v:.4f}" for k, v in report.items()))
    for mod, m in gm.items():
        print(f"[{fname}] grad {mod}: cos {m['cos']:.5f} rel {m['rel']:.4f} norm_err {m['norm_err']:.4f} worst tensor {m['worst'][0]:.4f} ({m['worst'][1]})")
    if sink == "bf16":
        return  # the packed-bf16 sink: recorded above, bounded per op in test_msda.py
    for k, v in report.items():  # measured: stages 2e-3 .. 7e-3, loss total < 1e-4, worst term 2e-2 (x 3-5 margins)
        bound = 2e-3 if k == "loss.total" else (5e-2 if k == "loss.worst_term" else 2e-2)
        assert v <= bound, (k, v, bound)
    for mod, m in gm.items():  # measured: cosine >= 0.99998, relative L2 <= 1.4e-2
        if mod != "decoder" and not seg:
            # detection: the encoder / backbone receive their gradient ONLY through the deformable attention's value gradient,
            # and in this fixture all 300 query boxes coincide (the reference's own top-k picks the zero-memory border tokens,
            # whose anchors are +inf: boxes = sigmoid(inf)), so that gradient is a sum of thousands of near-cancelling terms
            # per memory token -- rounding ``grad_out`` to bf16 alone moves the net by tens of percent (measured: cosine 0.61 /
            # 0.98 with float atomics, -0.5 / 0.92 with packed bf16 atomics). Recorded above, not asserted; the same kernels
            # are pinned by the segment fixture below, where the mask branch dominates those gradients, and per op in
            # test_msda.py.
            continue
        assert m["cos"] >= 0.999, (mod, m)
        assert m["rel"] <= 4e-2, (mod, m)


