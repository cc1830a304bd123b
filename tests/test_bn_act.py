"""Fused BatchNorm(+addend)+activation(+LAB) HIP kernels against a plain PyTorch fp32 reference
(F.batch_norm in training mode -> act -> scale*x+bias on the CPU), forward, all gradients and the
running-statistics update."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACTS = {None: lambda t: t, "relu": F.relu, "silu": F.silu}


def _reference(x, addend, gamma, beta, rm, rv, act, lab_scale, lab_bias):
    z = F.batch_norm(x, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    if addend is not None:
        z = z + addend
    y = ACTS[act](z)
    if lab_scale is not None:
        y = lab_scale * y + lab_bias
    return y


@pytest.mark.parametrize("B,C,H,W", [(2, 32, 160, 160), (3, 96, 20, 20), (2, 1536, 5, 5), (1, 8, 3, 7), (4, 256, 40, 40), (2, 21, 20, 20), (2, 12, 33, 31), (2, 42, 10, 10)])
@pytest.mark.parametrize("act,with_lab,with_add", [("relu", True, False), ("silu", False, False), (None, False, False), ("silu", False, True), ("relu", True, True)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_act_matches_torch_fp32(hip_lib, B, C, H, W, act, with_lab, with_add, dtype):
    from dfine_seg_amd import ops

    g = torch.Generator().manual_seed(C + H)
    # mean far from 0 on purpose: the statistics must survive |mean| >> std
    x = (torch.randn(B, C, H, W, generator=g) * 0.5 + 3.0).to(dtype)
    addend = torch.randn(B, C, H, W, generator=g).to(dtype) if with_add else None
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    ls, lb = (torch.tensor([1.1]), torch.tensor([-0.2])) if with_lab else (None, None)
    gy = torch.randn(B, C, H, W, generator=g).to(dtype)

    leaf = lambda t: t.float().clone().requires_grad_(True) if t is not None else None
    xr, ar, gr, br, lsr, lbr = leaf(x), leaf(addend), leaf(gamma), leaf(beta), leaf(ls), leaf(lb)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y_ref = _reference(xr, ar, gr, br, rm_ref, rv_ref, act, lsr, lbr)
    y_ref.backward(gy.float())

    dev = lambda t: t.cuda().requires_grad_(True) if t is not None else None
    cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) if t is not None else None
    xd, ad, gd, bd, lsd, lbd = cl(x), cl(addend), dev(gamma), dev(beta), dev(ls), dev(lb)
    rm_d, rv_d = rm.cuda(), rv.cuda()
    y = ops.batch_norm_act(xd, gd, bd, rm_d, rv_d, act, lsd, lbd, ad, training=True)
    assert y.dtype == dtype and y.shape == y_ref.shape
    y.backward(gy.cuda())

    if dtype == torch.float32:
        tol = dict(rtol=2e-4, atol=2e-4)
    else:  # one bf16 rounding of y / dx (rel 2^-9) on O(1..4) values
        tol = dict(rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(y.float().cpu(), y_ref.detach(), **tol)
    torch.testing.assert_close(rm_d.cpu(), rm_ref, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rv_d.cpu(), rv_ref, rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(xd.grad.float().cpu(), xr.grad, **tol)
    if with_add:
        torch.testing.assert_close(ad.grad.float().cpu(), ar.grad, **tol)
    n = B * H * W
    red_tol = dict(rtol=5e-3, atol=(2e-5 if dtype == torch.float32 else 2e-3) * n**0.5 + 1e-4)
    torch.testing.assert_close(gd.grad.cpu(), gr.grad, **red_tol)
    torch.testing.assert_close(bd.grad.cpu(), br.grad, **red_tol)
    if with_lab:
        lab_tol = dict(rtol=5e-3, atol=(2e-5 if dtype == torch.float32 else 2e-3) * (n * C) ** 0.5 + 1e-3)
        torch.testing.assert_close(lsd.grad.cpu(), lsr.grad, **lab_tol)
        torch.testing.assert_close(lbd.grad.cpu(), lbr.grad, **lab_tol)


def test_eval_mode_uses_running_statistics(hip_lib):
    from dfine_seg_amd import ops

    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 64, 9, 9, generator=g)
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    rm, rv = torch.randn(64, generator=g), torch.rand(64, generator=g) + 0.5
    ref = F.silu(F.batch_norm(x, rm, rv, gamma, beta, training=False, eps=1e-5))
    with torch.no_grad():
        y = ops.batch_norm_act(x.cuda().contiguous(memory_format=torch.channels_last), gamma.cuda(), beta.cuda(),
                               rm.cuda(), rv.cuda(), "silu", training=False)
    torch.testing.assert_close(y.cpu(), ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("frozen", [True, False])
@pytest.mark.parametrize("act,with_add", [("relu", False), ("silu", True), (None, False)])
def test_running_statistics_backward(hip_lib, frozen, act, with_add):
    """FrozenBatchNorm2d (buffers: only dx / d addend) and eval-mode BatchNorm with trainable affine
    (dgamma / dbeta too) through the fused kernels, against autograd of the same formula."""
    from dfine_seg_amd import ops

    g = torch.Generator().manual_seed(5)
    C = 40
    x = torch.randn(3, C, 11, 7, generator=g)
    a = torch.randn(3, C, 11, 7, generator=g) if with_add else None
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    gy = torch.randn(3, C, 11, 7, generator=g)
    xr = x.clone().requires_grad_(True)
    ar = a.clone().requires_grad_(True) if with_add else None
    gr, br = gamma.clone().requires_grad_(not frozen), beta.clone().requires_grad_(not frozen)
    z = F.batch_norm(xr, rm, rv, gr, br, training=False, eps=1e-5)
    if with_add:
        z = z + ar
    ref = {"relu": F.relu, "silu": F.silu, None: lambda t: t}[act](z)
    ref.backward(gy)
    cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last)
    xd = cl(x).requires_grad_(True)
    ad = cl(a).requires_grad_(True) if with_add else None
    gd, bd = gamma.cuda().requires_grad_(not frozen), beta.cuda().requires_grad_(not frozen)
    y = ops.batch_norm_act(xd, gd, bd, rm.cuda(), rv.cuda(), act, None, None, ad, training=False)
    torch.testing.assert_close(y.cpu(), ref.detach(), rtol=1e-4, atol=1e-4)
    y.backward(cl(gy))
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-4)
    if with_add:
        torch.testing.assert_close(ad.grad.cpu(), ar.grad, rtol=1e-4, atol=1e-4)
    if not frozen:
        torch.testing.assert_close(gd.grad.cpu(), gr.grad, rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(bd.grad.cpu(), br.grad, rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("B,C,H,W", [(2, 128, 40, 40), (3, 64, 9, 7), (2, 24, 16, 16), (1, 21, 10, 10)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bn_pair_act_matches_torch_fp32(hip_lib, B, C, H, W, dtype):
    """y = silu(bn_a(xa) + bn_b(xb)) in one launch (dfine_bn_pair_act_*, RepVgg's training form, hybrid_encoder.py:115-121)
    against F.batch_norm x 2 + silu in fp32: output, both input gradients, the four parameter gradients and the running
    statistics of both norms. The batch statistics arrive the way the model delivers them: in the workspaces the
    convolutions' epilogues fill (here filled by the standalone statistics pass)."""
    import torch.nn as nn

    from dfine_seg_amd import _lib, ops

    g = torch.Generator().manual_seed(C * 3 + H)
    xa = (torch.randn(B, C, H, W, generator=g) * 0.5 + 2.0).to(dtype)
    xb = (torch.randn(B, C, H, W, generator=g) * 1.5 - 1.0).to(dtype)
    gy = torch.randn(B, C, H, W, generator=g).to(dtype)
    norms_ref, norms = [], []
    for i in range(2):
        n = nn.BatchNorm2d(C)
        with torch.no_grad():
            n.weight.copy_(torch.rand(C, generator=g) + 0.5)
            n.bias.copy_(torch.randn(C, generator=g) * 0.1)
            n.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
            n.running_var.copy_(torch.rand(C, generator=g) + 0.5)
        import copy
        norms_ref.append(n.train())
        norms.append(copy.deepcopy(n).cuda().train())
    xar, xbr = xa.float().clone().requires_grad_(True), xb.float().clone().requires_grad_(True)
    y_ref = F.silu(norms_ref[0](xar) + norms_ref[1](xbr))
    y_ref.backward(gy.float())

    cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xad, xbd = cl(xa), cl(xb)
    ws = []
    for x in (xad, xbd):  # batch statistics into the norm workspaces, as a conv epilogue would leave them
        w = torch.zeros(34 * C, device="cuda")
        xl = x.detach().permute(0, 2, 3, 1).contiguous()
        dummy = torch.empty_like(xl)
        ones, zeros = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
        rc = hip_lib.dfine_bn_act_fwd(_lib.ptr(xl), None, _lib.ptr(dummy), _lib.dtype_code(xl), _lib.ptr(w), _lib.ptr(ones), _lib.ptr(zeros),
                                      _lib.ptr(zeros.clone()), _lib.ptr(ones.clone()), _lib.ptr(w[32 * C:]), _lib.ptr(w[33 * C:]), None, None,
                                      B * H * W, C, 0, 1, 1e-5, 0.1, 1, 0, None, 0, _lib.stream_ptr())
        assert rc == 0
        ws.append(w)
    assert ops.batch_norm_pair_act_supported(xad, xbd, norms[0], norms[1], ws[0], ws[1])
    y = ops.batch_norm_pair_act(xad, xbd, norms[0], norms[1], "silu", ws[0], ws[1])
    assert y.dtype == dtype and y.shape == y_ref.shape
    y.backward(gy.cuda())
    torch.cuda.synchronize()
    tol = dict(rtol=2e-4, atol=2e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(y.float().cpu(), y_ref.detach(), **tol)
    torch.testing.assert_close(xad.grad.float().cpu(), xar.grad, **tol)
    torch.testing.assert_close(xbd.grad.float().cpu(), xbr.grad, **tol)
    ptol = dict(rtol=2e-3, atol=2e-3 * (B * H * W) ** 0.5) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2 * (B * H * W) ** 0.5)
    for n, nr in zip(norms, norms_ref):
        torch.testing.assert_close(n.weight.grad.cpu(), nr.weight.grad, **ptol)
        torch.testing.assert_close(n.bias.grad.cpu(), nr.bias.grad, **ptol)
        torch.testing.assert_close(n.running_mean.cpu(), nr.running_mean, rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(n.running_var.cpu(), nr.running_var, rtol=1e-3, atol=1e-5)


def _bn_conv_chain(fused, B, C, H, W, K, ksize, act, with_lab, with_slot, branch):
    """x -> fused norm (+act, +LAB, optionally into a concat slice that a second reader also takes) -> stride-1 dense conv;
    returns (dx, dgamma, dbeta, dlab_scale, dlab_bias, dw, handle token)."""
    from dfine_seg_amd import ops

    torch.manual_seed(11)
    dev = "cuda"
    x = (torch.randn(B, C, H, W, device=dev) * 1.5 + 0.3).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_()
    gamma = (torch.rand(C, device=dev) + 0.5).requires_grad_()
    beta = (torch.randn(C, device=dev) * 0.2).requires_grad_()
    ls = torch.tensor([1.3], device=dev, requires_grad=True) if with_lab else None
    lb = torch.tensor([-0.1], device=dev, requires_grad=True) if with_lab else None
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    w = (torch.randn(K, C, ksize, ksize, device=dev) / (C * ksize * ksize) ** 0.5).contiguous(memory_format=torch.channels_last).requires_grad_()
    w1 = (torch.randn(K, C, 1, 1, device=dev) / C ** 0.5).contiguous(memory_format=torch.channels_last).requires_grad_()
    r = torch.randn(B, K, H, W, device=dev)
    old = ops.BNRED_FUSION
    ops.BNRED_FUSION = fused
    try:
        slot, buf = None, None
        if with_slot:
            buf = ops.ConcatBuffer(B, H, W, C + 16, torch.bfloat16, dev)
            slot = (buf, 8)
        y = ops.batch_norm_act(x, gamma, beta, rm, rv, act=act, lab_scale=ls, lab_bias=lb, training=True, slot=slot)
        handle = getattr(y, "_bn_handle", None)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if branch == "pair":  # RepVgg's 3x3 + 1x1 of one input: the SECOND (accumulating) launch carries the sums
                za, zb = ops.dense_conv_pair(y, w, w1, ksize // 2, 0)
                z = za + 0.5 * zb
            else:
                z = ops.dense_conv(y, w, 1, ksize // 2)
        loss = (z.float() * r).sum()
        if branch == "second_reader":  # autograd sums two gradients: the fused sums must be discarded
            loss = loss + (y.float() * 0.25).sum()
        if with_slot:  # the concat's reader leaves the slice's gradient in the buffer
            buf.grad = torch.randn(B, H, W, C + 16, device=dev).to(torch.bfloat16)
        loss.backward()
    finally:
        ops.BNRED_FUSION = old
    return (x.grad.float(), gamma.grad, beta.grad, ls.grad if with_lab else None, lb.grad if with_lab else None, w.grad.float(),
            handle.token if handle is not None else None)


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,H,W,K,ksize", [(2, 64, 40, 40, 96, 3), (3, 128, 20, 20, 128, 1), (2, 32, 37, 29, 48, 3), (1, 256, 10, 10, 64, 1),
                                             (4, 96, 80, 80, 96, 3)])
@pytest.mark.parametrize("act,with_lab,with_slot", [("relu", True, False), ("silu", False, False), (None, False, True), ("relu", True, True)])
def test_bn_backward_sums_from_the_consumer_conv_epilogue(hip_lib, B, C, H, W, K, ksize, act, with_lab, with_slot):
    """dfine_igemm_bf16_bnred: the norm's backward sums taken in the consumer conv's input-gradient epilogue give the same
    gradients as the norm's own reduction pass (same bf16 inputs, other summation order: 2e-3 relative L2 on dx, 1e-3 on the
    parameter gradients)."""
    got = _bn_conv_chain(True, B, C, H, W, K, ksize, act, with_lab, with_slot, None)
    ref = _bn_conv_chain(False, B, C, H, W, K, ksize, act, with_lab, with_slot, None)
    assert got[6] is not None and got[6][0] == "done", got[6]
    assert ref[6] is None
    for name, a, b, tol in zip(("dx", "dgamma", "dbeta", "dlab_scale", "dlab_bias", "dw"), got[:6], ref[:6], (2e-3, 1e-3, 1e-3, 1e-3, 1e-3, 1e-4)):
        if a is None:
            assert b is None
            continue
        rel = float((a - b).norm() / b.norm().clamp_min(1e-12))
        assert rel <= tol, (name, rel)


@pytest.mark.gpu
@pytest.mark.parametrize("with_slot", [False, True])
def test_bn_backward_sums_from_a_conv_pair_epilogue(hip_lib, with_slot):
    got = _bn_conv_chain(True, 2, 128, 20, 20, 128, 3, "silu", False, with_slot, "pair")
    ref = _bn_conv_chain(False, 2, 128, 20, 20, 128, 3, "silu", False, with_slot, "pair")
    assert got[6] is not None and got[6][0] == "done", got[6]
    for name, a, b, tol in zip(("dx", "dgamma", "dbeta"), got[:3], ref[:3], (2e-3, 1e-3, 1e-3)):
        rel = float((a - b).norm() / b.norm().clamp_min(1e-12))
        assert rel <= tol, (name, rel)


@pytest.mark.gpu
def test_bn_backward_discards_epilogue_sums_when_autograd_added_a_second_gradient(hip_lib):
    got = _bn_conv_chain(True, 2, 64, 24, 24, 64, 3, "relu", True, False, "second_reader")
    ref = _bn_conv_chain(False, 2, 64, 24, 24, 64, 3, "relu", True, False, "second_reader")
    assert got[6][0] == "done"  # the launch summed, the norm's backward saw another tensor and reduced by itself
    for name, a, b in zip(("dx", "dgamma", "dbeta", "dlab_scale", "dlab_bias"), got[:5], ref[:5]):
        rel = float((a - b).norm() / b.norm().clamp_min(1e-12))
        assert rel <= 1e-5, (name, rel)
