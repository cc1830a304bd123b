"""The product model + criterion on the MI355X against outputs of the reference
(tests/golden/model_*.npz): same name-hashed weights, same synthetic batch, the reference's
query selection (top-k indices) and denoising noise injected so per-index comparison is
meaningful (SURVEY.md 7 "Hard parts").

fp32 run: tolerance = kernel/op-order noise of an fp32 GPU pipeline vs fp32 CPU.
bf16 run (autocast, as trained): per-tensor relative L2 <= the bounds written below."""
import os

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


@pytest.mark.parametrize("fname", ["model_n320", "model_s640", "model_m640"])
def test_own_query_selection_equals_the_reference_topk_as_a_set(hip_lib, fname):
    """Index work is exact: with NOTHING injected for the query selection, the product's own top-300 of the 8 400 (500 for n)
    encoder tokens (decoder.py: ranking GEMMs + row-max + topk in fp32) is the reference's `topk_ind` as a SET per image
    (dfine_decoder.py:887-922; order inside the set may differ where two scores tie to round-off). Scores of the two fp32
    pipelines differ by ~1e-5 relative, so a swap can only happen between the 300th and 301st token if they tie that closely:
    at most one index per image may differ, and none does on these fixtures."""
    g, model, _, x, targets, inject, _ = _setup(fname)
    model.train()
    with torch.no_grad():
        res = model(x, targets=targets, inject={k: v for k, v in inject.items() if k != "topk_ind"})
    got, ref = res["enc_topk_ind"].cpu(), torch.as_tensor(g["topk_ind"])
    assert got.shape == ref.shape and got.dtype == torch.int64
    for b in range(ref.shape[0]):
        a, r = set(got[b].tolist()), set(ref[b].tolist())
        assert len(a) == ref.shape[1]
        assert len(a ^ r) <= 2, (fname, b, sorted(a ^ r))


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


class _StockOps:
    """Context manager: every fused / HIP op of ``dfine_seg_amd.ops`` declines or is replaced by the stock PyTorch op, so the
    SAME modules run as plain PyTorch (F.conv2d, nn.BatchNorm2d, F.relu, LAB, F.linear, SDPA, F.layer_norm) -- under
    ``torch.autocast`` this is the reference's AMP pipeline on this GPU, the yardstick for the product's bf16 path."""

    def __enter__(self):
        import torch.nn.functional as F

        from dfine_seg_amd import ops

        names = ("batch_norm_act_supported", "dense_conv", "depthwise_conv2d", "stem_pool_supported", "add_layer_norm_supported",
                 "linear", "self_attention_supported", "embedding", "conv_transpose2x2", "concat_buffer_supported",
                 "dense_conv_pair", "bilinear_add")
        self.ops, self.saved = ops, {n: getattr(ops, n) for n in names}
        ops.batch_norm_act_supported = lambda x, c: False
        ops.concat_buffer_supported = lambda x, parts: False  # torch.cat
        ops.dense_conv_pair = lambda x, wa, wb, pa, pb, sa=None, sb=None: (F.conv2d(x, wa, None, 1, pa), F.conv2d(x, wb, None, 1, pb))
        ops.dense_conv = lambda x, w, stride=1, padding=0, pad_br=None, stats=None, out=None: F.conv2d(
            F.pad(x, (0, pad_br, 0, pad_br)) if pad_br else x, w, None, stride, padding)
        ops.depthwise_conv2d = lambda x, w, stride=1, stats=None: F.conv2d(x, w, None, stride, (w.shape[-1] - 1) // 2, 1, x.shape[1])
        ops.stem_pool_supported = lambda x: False
        ops.add_layer_norm_supported = lambda x, d: False
        ops.linear = lambda x, w, b=None, r0=0, r1=None, relu=False: (F.relu if relu else (lambda t: t))(
            F.linear(x, w[r0:r1], b[r0:r1] if b is not None else None))
        ops.self_attention_supported = lambda qk, v, h: False
        ops.embedding = lambda idx, w, padding_idx=None: F.embedding(idx, w, padding_idx)
        ops.conv_transpose2x2 = lambda x, w, stats=None: F.conv_transpose2d(x, w, None, 2)
        ops.bilinear_add = lambda base, x: base + F.interpolate(x, size=base.shape[-2:], mode="bilinear", align_corners=False)
        return self

    def __exit__(self, *exc):
        for n, f in self.saved.items():
            setattr(self.ops, n, f)
        return False


def _stage_errors(stock):
    """Relative L2 distance to fp32 of the three backbone maps and of the three encoder maps (the encoder alone, fed the fp32
    features), for the product's bf16 path or for the same modules as stock PyTorch bf16 autocast: ([3], [3])."""
    g, model, crit, x, targets, inject, seg = _setup("model_m640")
    model.train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    xcl = x.contiguous(memory_format=torch.channels_last)

    def stages(autocast):
        model.load_state_dict(state)  # identical BatchNorm running statistics for every run
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            return model.backbone(xcl)

    def encoder_from(feats, autocast):
        model.load_state_dict(state)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            outs, _ = model.encoder([f.clone() for f in feats])
        return outs

    feats32 = stages(False)
    outs32 = encoder_from(feats32, False)
    if stock:
        with _StockOps():
            feats16, outs16 = stages(True), encoder_from(feats32, True)
    else:
        feats16, outs16 = stages(True), encoder_from(feats32, True)
    torch.cuda.synchronize()
    return ([_rel(feats16[l].float().cpu(), feats32[l].float().cpu()) for l in range(3)],
            [_rel(outs16[l].float().cpu(), outs32[l].float().cpu()) for l in range(3)])


def _stock_stage_errors_worker(path):
    torch.save(_stage_errors(True), path)


def _in_child(worker, path, timeout=900):
    """Run ``worker(path)`` in a spawned process (the stock MIOpen pipeline can take the process down, see below); the loaded
    result, or None if the child died."""
    import multiprocessing as mp

    child = mp.get_context("spawn").Process(target=worker, args=(path,))
    child.start()
    child.join(timeout)
    if child.is_alive():
        child.kill()
        child.join()
    return torch.load(path, weights_only=True) if child.exitcode == 0 and os.path.exists(path) else None


def test_bf16_stage_errors_are_bounded_and_below_stock_autocast(hip_lib, tmp_path):
    """Stage-wise bf16 accuracy of the path that is trained and benchmarked (D-FINE-m, name-hashed weights, the 2-image golden
    batch; train-mode BatchNorm): each stage is fed the SAME fp32-computed input and its bf16 output is compared with its
    fp32 output (whose end-to-end agreement with the reference is pinned at 2e-4 above), next to the same modules run as stock
    PyTorch bf16 autocast. With random weights and a 2-image batch, round-off grows ~7 % per BatchNorm layer in ANY bf16
    pipeline (profiles/r02_bf16_error_by_layer.txt: stock autocast reaches 1.08 at the end of the backbone and 1.2-1.36 in
    the encoder; the reference's own 0.27-0.41 in model_m640.npz is its CPU autocast, which keeps BatchNorm outputs in fp32).
    The product rounds once per layer (conv -> fused norm+act+LAB in fp32 -> bf16) instead of after every op and must stay
    BELOW the stock pipeline at every stage, and within the absolute bounds written here (measured x 1.5). The stock pipeline
    runs in a child process (see test_bf16_train_step_gradients_vs_fp32_and_stock_autocast); without it only the absolute
    bounds are enforced."""
    prod_b, prod_e = _stage_errors(False)
    stock = _in_child(_stock_stage_errors_worker, str(tmp_path / "stock_stage_errors.pt"))
    bounds_backbone, bounds_encoder = (0.12, 0.30, 0.45), (0.75, 0.90, 1.05)
    for lvl in range(3):
        assert prod_b[lvl] < bounds_backbone[lvl], ("backbone", lvl, prod_b[lvl])
        assert prod_e[lvl] < bounds_encoder[lvl], ("encoder", lvl, prod_e[lvl])
        if stock is not None:
            assert prod_b[lvl] < 0.6 * stock[0][lvl], ("backbone", lvl, prod_b[lvl], stock[0][lvl])
            assert prod_e[lvl] < stock[1][lvl], ("encoder", lvl, prod_e[lvl], stock[1][lvl])
    if stock is None:
        pytest.xfail("the stock bf16 pipeline died in its child process (MIOpen igemm_bwd_gtcx35_nhwc_* over-read, see "
                     "tests/conftest.py): only the absolute bounds were checked")


def _step_gradients(amp, stock):
    """Gradients of one D-FINE-s training step (name-hashed weights, golden batch, injected noise / query selection) through
    TrainStep's forward + backward: fp32 (amp=False), the product's bf16 path, or the same modules as stock PyTorch bf16."""
    from dfine_seg_amd.targets import TargetBatch
    from dfine_seg_amd.train import TrainStep

    size, img, n_gts, seg = MODEL_CASES["model_s640"]
    gold = load_golden("model_s640")
    x = weights.synthetic_images(len(n_gts), img, img, SEED_WEIGHTS_IMG).cuda()
    targets = weights.synthetic_targets(n_gts, 80, SEED_TARGETS, with_masks=False, height=img, width=img)
    torch.manual_seed(SEED_CDN)
    noise = oracle_ops.cdn_noise(n_gts, 80)
    inject = {"topk_ind": torch.as_tensor(gold["topk_ind"]).cuda(), "cdn_noise": {k: v.cuda() for k, v in noise.items()}}
    t = TrainStep(size, 80, False, (img, img), "cuda", amp=amp, use_graph=False)
    sd = weights.fill_state_dict(t.model.state_dict())
    t.model.load_state_dict(sd, strict=True)
    t.flat.refresh_bf16_shadow()
    tb = TargetBatch.from_list(targets, 80, t.device)
    if stock:
        with _StockOps():
            t._forward_backward(x, tb, inject)
    else:
        t._forward_backward(x, tb, inject)
    torch.cuda.synchronize()
    return {n: p.grad.detach().float().cpu().clone() for n, p in t.model.named_parameters() if p.requires_grad}


def _stock_gradients_worker(path):
    torch.save(_step_gradients(True, True), path)


def test_bf16_train_step_gradients_vs_fp32_and_stock_autocast(hip_lib, tmp_path):
    """The gradients the benchmarked path feeds its optimizer -- bf16 shadow weights, implicit-GEMM / attention kernels, the
    wgrad kernels accumulating straight into the flat fp32 gradient buffer (TrainStep, amp=True) -- against (a) the fp32
    gradients of the same step (pinned to the reference by grad norms above) and (b) the gradients stock PyTorch bf16 autocast
    + autograd produces for the same modules. D-FINE-s, reference noise / query selection injected. Bound: cosine similarity
    with the fp32 gradient >= the stock pipeline's minus 0.03 and >= 0.5, for the whole model and for each of backbone /
    encoder / decoder. (Measured, whole model: product 0.65, stock autocast 0.59 -- with random weights and BatchNorm over 2
    images NO bf16 pipeline tracks the fp32 gradient closely, see profiles/r02_bf16_error_by_layer.txt; the product must not be
    the worse of the two.)
    The STOCK pipeline runs in a child process: MIOpen's assembly implicit-GEMM NHWC backward-data kernel
    (igemm_bwd_gtcx35_nhwc_*) reads past the end of an operand and faults when that operand ends an allocator segment
    (cause and evidence: tests/conftest.py, profiles/r03_fault_diag_launch_tail.txt); that solver is now switched off for
    the test processes, the child process stays as a second line of defence for the yardstick. If the child dies the
    comparison (b) is reported as xfail and only the absolute bound (a) is enforced."""
    g32, g_prod = _step_gradients(False, False), _step_gradients(True, False)
    g_stock = _in_child(_stock_gradients_worker, str(tmp_path / "stock_grads.pt"))

    def cos(a, b, prefix=""):
        names = [n for n in g32 if n.startswith(prefix)]
        va, vb = torch.cat([a[n].flatten() for n in names]), torch.cat([b[n].flatten() for n in names])
        return float(torch.dot(va, vb) / (va.norm() * vb.norm()))

    for prefix in ("", "backbone.", "encoder.", "decoder."):
        c_prod = cos(g_prod, g32, prefix)
        assert c_prod >= 0.5, (prefix, c_prod)
        if g_stock is not None:
            c_stock = cos(g_stock, g32, prefix)
            assert c_prod >= c_stock - 0.03, (prefix, c_prod, c_stock)
    if g_stock is None:
        pytest.xfail("the stock bf16 pipeline died in its child process (MIOpen igemm_bwd_gtcx35_nhwc_* over-read, see "
                     "tests/conftest.py): only the fp32 bound was checked")


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
    BN, 16 % leaving it): the encoder maps are bounded stage by stage, against fp32 AND against stock PyTorch bf16
    autocast on this GPU, in test_bf16_stage_errors_are_bounded_and_below_stock_autocast; here they only have to stay
    correlated with fp32 end to end, and the encoder class logits within 2x of the reference's own bf16 distance. The tensors that
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
    for lvl, f in enumerate(grabbed["outs"]):  # end-to-end distance to fp32 (bounded stage by stage in the test above)
        assert _rel(_sample(f)[0], g[f"encoder.outs.{lvl}.sample"]) < 1.1, lvl
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


def test_x960_full_size_step_properties(hip_lib):
    """BASELINE config 5 at its real size (D-FINE-x, 960x960: 18 900 memory tokens, 900 AIFI tokens with 48-wide heads, 384-wide
    encoder with the 384 -> 256 decoder projection, frozen backbone BatchNorm, 6 decoder layers), batch 2, through the full
    bf16 training step and the eval forward -- size-independent properties instead of a golden (the x fixture is 640x640):
    finite loss terms with the reference's 68 keys, finite weights after the fused optimizer, gradients zeroed, the 384-wide
    LayerNorm on the HIP kernel, eval outputs finite, boxes inside the image, and two eval runs bit-identical."""
    from dfine_seg_amd import ops
    from dfine_seg_amd.synth import synthetic_batch
    from dfine_seg_amd.train import TrainStep

    torch.manual_seed(0)
    t = TrainStep("x", 80, False, (960, 960), "cuda", amp=True, use_graph=False, total_steps=100)
    images, targets = synthetic_batch(2, 960, 960, 80, 8, seed=3, device="cuda")
    assert ops.add_layer_norm_supported(torch.empty(2, 900, 384, device="cuda", dtype=torch.bfloat16), 384)
    for _ in range(2):
        loss = t.step(images, targets)
    assert torch.isfinite(loss)
    assert len(t.last_losses) == 68 and all(torch.isfinite(v) for v in t.last_losses.values())
    assert torch.isfinite(t.flat.params).all() and float(t.flat.grads.abs().max()) == 0.0
    assert torch.equal(t.flat.shadow, t.flat.params.to(torch.bfloat16))
    t.model.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        a = t.model(images)
        b = t.model(images)
    assert a["pred_logits"].shape == (2, 300, 80) and a["pred_boxes"].shape == (2, 300, 4)
    assert torch.isfinite(a["pred_logits"].float()).all() and torch.isfinite(a["pred_boxes"]).all()
    assert float(a["pred_boxes"].min()) >= -1e-3 and float(a["pred_boxes"].max()) <= 1.5
    assert torch.equal(a["pred_logits"], b["pred_logits"]) and torch.equal(a["pred_boxes"], b["pred_boxes"])


def test_mask_heads_batched_into_one_gemm_match_the_per_head_form(hip_lib, monkeypatch):
    """criterion._batched_mask_stats: the matched rows of every head x the shared mask feature map as ONE GEMM + one statistics
    launch, against the per-head form (one einsum per head, autograd summing the feature-map gradients), same fp32 model,
    same inputs: loss terms to 1e-5, parameter gradients to 1e-4 (summation order of the feature-map gradient)."""
    from dfine_seg_amd import criterion as crit_mod

    def run(batched):
        g, model, crit, x, targets, inject, seg = _setup("model_m640_seg")
        assert seg
        model.train()
        monkeypatch.setattr(crit_mod, "BATCH_MASK_HEADS", batched)
        losses = crit(model(x, targets=targets, inject=inject), targets)
        sum(losses.values()).backward()
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        return {k: float(v) for k, v in losses.items()}, grads

    l1, g1 = run(True)
    l0, g0 = run(False)
    assert list(l1) == list(l0)
    for k in l0:
        assert abs(l1[k] - l0[k]) <= 1e-5 * max(1.0, abs(l0[k])), (k, l1[k], l0[k])
    assert g1.keys() == g0.keys()
    # scale-invariant parameters (a norm weight in front of another norm) have a true gradient of 0 and hold round-off only:
    # the absolute floor, relative to the largest gradient of the model, covers them
    gmax = max(float(v.norm()) for v in g0.values())
    for n in g0:
        diff = float((g1[n] - g0[n]).norm())
        # two runs of ONE formulation already differ by ~4e-3 at the stem (float atomics in the weight gradients, amplified
        # through the random-weight net): the global bound only guards against a wrong term
        assert diff <= 2e-2 * float(g0[n].norm()) + 1e-4 * gmax, (n, diff, float(g0[n].norm()), gmax)
    for n in ("decoder.mask_head.layers.2.weight", "decoder.pixel_decoder.upconv.weight", "decoder.pixel_decoder.lateral.0.weight"):
        assert _rel(g1[n], g0[n].cpu()) < 5e-3, n  # the parameters right behind the batched GEMM (run-to-run noise: ~1e-3)
