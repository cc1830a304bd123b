"""ops.ConcatBuffer: the HG block without torch.cat. Producers write channel slices of one buffer (fused norm kernel with an
output pitch), consumers read slices in place, and in backward every producer adds its slice of the concat's gradient on load
(dfine_bn_act_bwd's dy2). Reference: the same block run through torch.cat + autograd's own gradient accumulation."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run_block(block, x, mode, monkeypatch):
    """mode: "buffer" (bf16, ConcatBuffer), "cat" (bf16, torch.cat + autograd's accumulation), "fp32" (no autocast: the truth)."""
    from dfine_seg_amd import ops

    if mode == "cat":
        monkeypatch.setattr(ops, "concat_buffer_supported", lambda *a, **k: False)
    xx = (x.float() if mode == "fp32" else x.clone()).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode != "fp32"):
        y = block(xx)
    w = torch.linspace(-1, 1, y.numel(), device="cuda").view_as(y)
    (y.float() * w).sum().backward()
    grads = {n: p.grad.clone() for n, p in block.named_parameters()}
    monkeypatch.undo()
    return y.detach(), xx.grad.detach(), grads


@pytest.mark.parametrize("light,k,residual", [(True, 5, True), (False, 3, False)])
def test_hg_block_without_cat_matches_the_cat_path(hip_lib, monkeypatch, light, k, residual):
    from dfine_seg_amd.backbone import Block

    torch.manual_seed(0)
    kw = dict(act="relu", lab=True, norm_name="bn", frozen_norm=False)
    c_in = 64
    block = Block(c_in, 32, 64, 3, k, light, residual, kw).cuda().to(memory_format=torch.channels_last).train()
    x = torch.randn(4, c_in, 24, 24, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    cat_block, fp32_block = copy.deepcopy(block), copy.deepcopy(block)
    y0, gx0, g0 = _run_block(cat_block, x, "cat", monkeypatch)
    y1, gx1, g1 = _run_block(block, x, "buffer", monkeypatch)
    _, gxr, gr = _run_block(fp32_block, x, "fp32", monkeypatch)
    # forward: the same kernels on the same values, only the addresses differ -- except the residual, which the buffer path's
    # last norm kernel adds before ITS single rounding (the cat path rounds y, then y + x)
    if residual:
        torch.testing.assert_close(y1.float(), y0.float(), rtol=8e-3, atol=1e-2)
    else:
        assert torch.equal(y0, y1)
    # running statistics were updated identically
    for (n, b0), (_, b1) in zip(cat_block.named_buffers(), block.named_buffers()):
        assert torch.equal(b0, b1), n
    # backward: a two-consumer gradient is summed in fp32 on load instead of being rounded to a bf16 tensor first, so the two
    # bf16 runs are a bf16 round-off apart; measured against the fp32 run the buffer path must be no worse than the cat path
    # (sums with heavy cancellation -- the LAB bias gradient is the sum of ALL gradient elements -- move by percents in both)
    rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))
    assert rel(gx1, gx0) < 1e-2, rel(gx1, gx0)
    assert rel(gx1, gxr) < max(1.25 * rel(gx0, gxr), 1e-2)
    for n in g0:
        if light and n.endswith("conv1.bn.weight"):
            # a per-channel scale in front of a depthwise conv + BatchNorm cannot change the output (the second norm divides
            # it out): the true gradient is 0 and every run holds nothing but round-off there
            assert float(g1[n].abs().max()) < 1.0
            continue
        if rel(g0[n], gr[n]) > 0.5:
            # the weights of the test loss sum to zero, so the true LAB bias gradient (the sum of ALL gradient elements) is ~0
            # and both bf16 runs hold only cancellation noise there, whose size also depends on the atomics' order: same
            # magnitude is all that can be asked
            assert float((g1[n].float() - gr[n].float()).norm()) < 4 * float((g0[n].float() - gr[n].float()).norm()), n
            continue
        assert rel(g1[n], gr[n]) < max(1.25 * rel(g0[n], gr[n]), 2e-2), (n, rel(g1[n], gr[n]), rel(g0[n], gr[n]))


def test_norm_kernel_writes_a_slice_and_sums_two_gradients(hip_lib):
    """dfine_bn_act_fwd with an output pitch and dfine_bn_act_bwd with (dy, dy2) against the dense call on dy + dy2."""
    from dfine_seg_amd import ops

    torch.manual_seed(1)
    B, C, H, W, CT, c0 = 3, 32, 10, 12, 96, 40
    x = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    gamma = (torch.rand(C, device="cuda") + 0.5).requires_grad_(True)
    beta = torch.randn(C, device="cuda").requires_grad_(True)
    g_next = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    g_cat = torch.randn(B, H, W, CT, device="cuda").bfloat16()

    def stats():
        return torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")

    # dense reference: one gradient = bf16(dy + dy2 slice) the way autograd would accumulate it... in fp32 here, so the
    # comparison isolates the kernels from that rounding
    xr = x.clone().requires_grad_(True)
    rm, rv = stats()
    y_ref = ops.batch_norm_act(xr, gamma, beta, rm, rv, "relu", training=True)
    total = (g_next.float() + g_cat[..., c0 : c0 + C].permute(0, 3, 1, 2).float())
    y_ref.backward(total.bfloat16())
    ref = (xr.grad.clone(), gamma.grad.clone(), beta.grad.clone())
    gamma.grad = beta.grad = None

    buf = ops.ConcatBuffer(B, H, W, CT, torch.bfloat16, "cuda")
    buf.buf.fill_(7.0)
    buf.used = c0
    xs = x.clone().requires_grad_(True)
    rm2, rv2 = stats()
    y = ops.batch_norm_act(xs, gamma, beta, rm2, rv2, "relu", training=True, slot=(buf, buf.take(C)))
    assert y.data_ptr() == buf.buf[..., c0:].data_ptr() and torch.equal(y, y_ref)
    assert float((buf.buf[..., :c0] - 7).abs().max()) == 0 and float((buf.buf[..., c0 + C :] - 7).abs().max()) == 0  # neighbours untouched
    assert torch.equal(rm, rm2) and torch.equal(rv, rv2)
    buf.grad = g_cat
    y.backward(g_next)
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    assert rel(xs.grad, ref[0]) < 6e-3, rel(xs.grad, ref[0])  # the reference rounded the summed gradient to bf16 first
    assert rel(gamma.grad, ref[1]) < 6e-3 and rel(beta.grad, ref[2]) < 6e-3
    # only the concat's gradient (the last layer of a block has no other consumer)
    xs.grad = gamma.grad = beta.grad = None
    rm3, rv3 = stats()
    buf2 = ops.ConcatBuffer(B, H, W, CT, torch.bfloat16, "cuda")
    buf2.used = c0
    x3 = x.clone().requires_grad_(True)
    y3 = ops.batch_norm_act(x3, gamma, beta, rm3, rv3, "relu", training=True, slot=(buf2, buf2.take(C)))
    buf2.used = CT
    full = ops.concat_read(buf2, [y3])
    full.backward(g_cat.permute(0, 3, 1, 2))
    x4 = x.clone().requires_grad_(True)
    g3 = (gamma.grad.clone(), beta.grad.clone())
    gamma.grad = beta.grad = None
    rm4, rv4 = stats()
    ops.batch_norm_act(x4, gamma, beta, rm4, rv4, "relu", training=True).backward(g_cat[..., c0 : c0 + C].permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last))
    assert torch.equal(x3.grad, x4.grad) and torch.allclose(g3[0], gamma.grad, rtol=1e-5, atol=1e-5)


def test_stem_without_cat_matches_the_cat_path(hip_lib, monkeypatch):
    """HGNetv2 stem: the max-pooled map (dfine_stem_pool_fwd with an output pitch) and the 2x2-conv branch fill the two halves
    of one buffer; in backward the pool reads its half of stem3's input gradient in place (ld_gy). Forward bit-equal to the
    torch.cat path; gradients no worse than the cat path against the fp32 run."""
    from dfine_seg_amd.backbone import Stem

    torch.manual_seed(0)
    kw = dict(act="relu", lab=True, norm_name="bn", frozen_norm=False)
    stem = Stem(3, 24, 32, kw).cuda().to(memory_format=torch.channels_last).train()
    x = torch.randn(2, 3, 64, 96, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    cat_stem, fp32_stem = copy.deepcopy(stem), copy.deepcopy(stem)
    y0, gx0, g0 = _run_block(cat_stem, x, "cat", monkeypatch)
    y1, gx1, g1 = _run_block(stem, x, "buffer", monkeypatch)
    _, gxr, gr = _run_block(fp32_stem, x, "fp32", monkeypatch)
    assert torch.equal(y0, y1)
    rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))
    assert rel(gx1, gxr) < max(1.25 * rel(gx0, gxr), 1e-2), (rel(gx1, gxr), rel(gx0, gxr))
    for n in g0:
        if rel(g0[n], gr[n]) > 0.5:
            # the weights of the test loss sum to zero, so the true LAB bias gradient (the sum of ALL gradient elements) is ~0
            # and both bf16 runs hold only cancellation noise there, whose size also depends on the atomics' order: same
            # magnitude is all that can be asked
            assert float((g1[n].float() - gr[n].float()).norm()) < 4 * float((g0[n].float() - gr[n].float()).norm()), n
            continue
        assert rel(g1[n], gr[n]) < max(1.25 * rel(g0[n], gr[n]), 2e-2), (n, rel(g1[n], gr[n]), rel(g0[n], gr[n]))


def test_encoder_fpn_pan_without_cat_matches_the_cat_path(hip_lib, monkeypatch):
    """HybridEncoder (D-FINE-s): the FPN's upsampled map (dfine_upsample2x_fwd with an output pitch), the lateral input
    projection and the PAN's downsampled map are written into the halves of each block's input buffer. Outputs bit-equal to
    the torch.cat / F.interpolate path; input and parameter gradients no worse than that path against the fp32 run."""
    from dfine_seg_amd.configs import MODELS
    from dfine_seg_amd.encoder import HybridEncoder

    torch.manual_seed(0)
    spec = MODELS["s"]
    enc = HybridEncoder(spec).cuda().to(memory_format=torch.channels_last).train()
    feats = [torch.randn(2, c, 64 // s, 96 // s, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
             for c, s in zip(spec.in_channels, (1, 2, 4))]

    def run(model, mode):
        from dfine_seg_amd import ops

        if mode == "cat":
            monkeypatch.setattr(ops, "concat_buffer_supported", lambda *a, **k: False)
        xs = [(f.float() if mode == "fp32" else f.clone()).requires_grad_(True) for f in feats]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode != "fp32"):
            outs, inner = model(xs)
        loss = sum((o.float() * torch.linspace(-1, 1, o.numel(), device="cuda").view_as(o)).sum() for o in outs)
        loss.backward()
        monkeypatch.undo()
        return [o.detach() for o in outs], [x.grad.detach() for x in xs], {n: p.grad.clone() for n, p in model.named_parameters()}

    cat_enc, fp32_enc = copy.deepcopy(enc), copy.deepcopy(enc)
    o0, gx0, g0 = run(cat_enc, "cat")
    o1, gx1, g1 = run(enc, "buffer")
    _, gxr, gr = run(fp32_enc, "fp32")
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))
    for a, b, r in zip(gx1, gx0, gxr):
        assert rel(a, r) < max(1.25 * rel(b, r), 2e-2), (rel(a, r), rel(b, r))
    worse = [n for n in g0 if rel(g1[n], gr[n]) > max(1.25 * rel(g0[n], gr[n]), 5e-2)]
    assert len(worse) <= len(g0) // 50, worse  # a handful of near-zero-gradient parameters are pure round-off in every run


def test_elan_block_without_cat_or_chunk_matches_the_cat_path(hip_lib, monkeypatch):
    """RepNCSPELAN4 (hybrid_encoder.py:213-239): cv1 and the two branches' last norms write their slices of one buffer, the first
    branch reads the second half of cv1's slice in place and adds its input gradient to the parked concat gradient
    (ops.concat_slot_slice). Forward bit-equal to chunk + torch.cat; gradients no worse than that path against the fp32 run."""
    from dfine_seg_amd.encoder import RepNCSPELAN4

    torch.manual_seed(0)
    block = RepNCSPELAN4(128, 64, 64, 32, 2).cuda().to(memory_format=torch.channels_last).train()
    x = torch.randn(3, 128, 20, 28, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    cat_block, fp32_block = copy.deepcopy(block), copy.deepcopy(block)
    y0, gx0, g0 = _run_block(cat_block, x, "cat", monkeypatch)
    y1, gx1, g1 = _run_block(block, x, "buffer", monkeypatch)
    _, gxr, gr = _run_block(fp32_block, x, "fp32", monkeypatch)
    assert torch.equal(y0, y1)
    for (n, b0), (_, b1) in zip(cat_block.named_buffers(), block.named_buffers()):
        assert torch.equal(b0, b1), n
    rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))
    assert rel(gx1, gxr) < max(1.25 * rel(gx0, gxr), 1e-2), (rel(gx1, gxr), rel(gx0, gxr))
    worse = [n for n in g0 if rel(g1[n], gr[n]) > max(1.25 * rel(g0[n], gr[n]), 5e-2)]
    assert len(worse) <= max(1, len(g0) // 25), [(n, rel(g1[n], gr[n]), rel(g0[n], gr[n])) for n in worse]


def test_norm_kernel_adds_a_residual_after_the_activation(hip_lib):
    """dfine_bn_act_fwd ``post``: y = lab(act(norm(x))) + post in one launch, the residual read with its own pitch; against
    the two-step form on the same inputs (one bf16 rounding less: within a bf16 ulp), and the residual's gradient = the output
    gradient."""
    from dfine_seg_amd import ops

    torch.manual_seed(5)
    B, C, H, W = 2, 48, 9, 11
    x = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    wide = torch.randn(B, H, W, 2 * C, device="cuda").bfloat16()
    gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    ls, lb = torch.tensor([1.3], device="cuda"), torch.tensor([-0.2], device="cuda")
    g = torch.randn(B, C, H, W, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    outs = []
    for fused in (True, False):
        xs = x.clone().requires_grad_(True)
        res = wide.clone().requires_grad_(True)
        post = res[..., C:].permute(0, 3, 1, 2)  # a channel slice of a wider map: pitch 2C
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        if fused:
            y = ops.batch_norm_act(xs, gamma, beta, rm, rv, "relu", ls, lb, training=True, post=post)
        else:
            y = ops.batch_norm_act(xs, gamma, beta, rm, rv, "relu", ls, lb, training=True) + post
        y.backward(g)
        outs.append((y.detach().float(), xs.grad.float(), res.grad.float()))
    (y1, gx1, gr1), (y0, gx0, gr0) = outs
    torch.testing.assert_close(y1, y0, rtol=8e-3, atol=8e-3)
    assert torch.equal(gx1, gx0) and torch.equal(gr1, gr0)
    assert float(gr1[..., :C].abs().max()) == 0 and torch.equal(gr1[..., C:], g.permute(0, 2, 3, 1).float())


def test_stage_chains_its_blocks_through_their_buffers(hip_lib, monkeypatch):
    """HG stage: the downsample's norm and every block's last norm write the NEXT block's first slice (no copy-in), the
    residual rides in that norm kernel. Against the unchained blocks (copy-in + add kernels) and the fp32 run."""
    from dfine_seg_amd import backbone
    from dfine_seg_amd.configs import BACKBONES

    torch.manual_seed(0)
    kw = dict(act="relu", lab=True, norm_name="bn", frozen_norm=False)
    spec = BACKBONES["B2"].stages[2]  # downsample, light 5x5 layers, residual blocks after the first
    stage = backbone.Stage(spec, kw).cuda().to(memory_format=torch.channels_last).train()
    assert len(stage.blocks) > 1 and hasattr(stage, "downsample")
    x = torch.randn(2, spec.c_in, 24, 20, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)

    def run(model, mode):
        monkeypatch.setattr(backbone, "CHAIN_BLOCKS", mode == "chained")
        xx = (x.float() if mode == "fp32" else x.clone()).requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode != "fp32"):
            y = model(xx)
        (y.float() * torch.linspace(-1, 1, y.numel(), device="cuda").view_as(y)).sum().backward()
        monkeypatch.undo()
        return y.detach(), xx.grad.detach(), {n: p.grad.clone() for n, p in model.named_parameters()}

    plain, fp32 = copy.deepcopy(stage), copy.deepcopy(stage)
    y0, gx0, g0 = run(plain, "plain")
    y1, gx1, g1 = run(stage, "chained")
    yr, gxr, gr = run(fp32, "fp32")
    rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))
    # the fused residual rounds once where (y -> bf16) + x rounds twice: not bit-equal, both a bf16 round-off from fp32
    assert rel(y1, yr) < max(1.25 * rel(y0, yr), 1e-2), (rel(y1, yr), rel(y0, yr))
    assert rel(gx1, gxr) < max(1.25 * rel(gx0, gxr), 2e-2), (rel(gx1, gxr), rel(gx0, gxr))
    worse = [n for n in g0 if rel(g1[n], gr[n]) > max(1.25 * rel(g0[n], gr[n]), 5e-2)]
    assert len(worse) <= max(1, len(g0) // 25), [(n, rel(g1[n], gr[n]), rel(g0[n], gr[n])) for n in worse]
