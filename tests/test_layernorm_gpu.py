"""Fused residual-add + LayerNorm (forward / backward) against F.layer_norm autograd in fp32."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,D,with_res,clamp", [(15744, 256, True, None), (1000, 256, False, None), (77, 128, True, 3.0),
                                                 (9600, 256, True, 65504.0), (5, 128, False, None), (300, 512, True, None),
                                                 (1800, 384, True, None)])  # 384: D-FINE-x AIFI
def test_add_layer_norm_matches_torch(hip_lib, dtype, M, D, with_res, clamp):
    from dfine_seg_amd import ops

    if D == 512 and dtype == torch.float32:
        pytest.skip("D=512 is built for bf16 only")
    g = torch.Generator(device="cuda").manual_seed(M + D)
    x = (torch.randn(M, D, device="cuda", generator=g) * 2).to(dtype)
    res = (torch.randn(M, D, device="cuda", generator=g) * 2).to(dtype) if with_res else None
    w = torch.randn(D, device="cuda", generator=g) * 0.5 + 1
    b = torch.randn(D, device="cuda", generator=g) * 0.1
    gy = torch.randn(M, D, device="cuda", generator=g).to(dtype)
    # reference: fp32 math on the same (rounded) inputs; the sum is rounded to dtype like the kernel's saved s
    xr = x.detach().float().requires_grad_(True)
    rr = res.detach().float().requires_grad_(True) if with_res else None
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    s = xr + rr if with_res else xr
    if clamp:
        s = s.clamp(-clamp, clamp)
    if dtype == torch.bfloat16 and (with_res or clamp):
        s = s + (s.detach().to(dtype).float() - s.detach())  # straight-through rounding
    F.layer_norm(s, (D,), wr, br, 1e-5).backward(gy.float())
    for direct in (False, True):
        xd = x.detach().clone().requires_grad_(True)
        rd = res.detach().clone().requires_grad_(True) if with_res else None
        wd, bd = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        if direct:
            wd.grad, bd.grad = torch.zeros_like(wd), torch.zeros_like(bd)
        y = ops.add_layer_norm(xd, rd, wd, bd, 1e-5, clamp)
        assert y.dtype == dtype
        ref_y = F.layer_norm(s.detach(), (D,), w, b, 1e-5)
        tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=3e-2)
        torch.testing.assert_close(y.float(), ref_y, **tol)
        y.backward(gy)
        # the backward tells saturation from the SAVED (rounded) sum: in bf16 an unsaturated 2.996 and a clamped
        # 3.0 are the same number -- leave those boundary elements out (the model's clamp is 65504: never hit)
        inside = torch.ones_like(xr, dtype=torch.bool) if not clamp else (s.detach().abs() < clamp)
        torch.testing.assert_close(xd.grad.float() * inside, xr.grad * inside, **tol)
        if with_res:
            torch.testing.assert_close(rd.grad.float() * inside, rr.grad * inside, **tol)
        wtol = dict(rtol=2e-3, atol=2e-3 * (M ** 0.5)) if dtype == torch.float32 else dict(rtol=2e-2, atol=3e-2 * (M ** 0.5))
        torch.testing.assert_close(wd.grad, wr.grad, **wtol)
        torch.testing.assert_close(bd.grad, br.grad, **wtol)


def test_add_layer_norm_unsupported_width(hip_lib):
    from dfine_seg_amd import _lib

    z = torch.zeros(4 * 320, device="cuda")
    assert hip_lib.dfine_add_layernorm_fwd(_lib.ptr(z), None, 0, _lib.ptr(z), _lib.ptr(z), _lib.ptr(z), None, _lib.ptr(z), _lib.ptr(z),
                                           4, 320, 1e-5, 0.0, None) == -2


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gate_mix_matches_the_op_by_op_gateway(hip_lib, dtype):
    """ops.gate_mix (dfine_gate_mix_fwd / _bwd) vs sigmoid(pre).chunk -> g1 * a + g2 * b with autograd, same dtype:
    fp32 to 1e-6; bf16 within one rounding of the op-by-op form (which rounds sigmoid, products and the sum to bf16 too)."""
    from dfine_seg_amd import ops

    torch.manual_seed(0)
    M, D = 1000, 256
    pre = torch.randn(4, M // 4, 2 * D, device="cuda").to(dtype).requires_grad_(True)
    a = torch.randn(4, M // 4, D, device="cuda").to(dtype).requires_grad_(True)
    b = torch.randn(4, M // 4, D, device="cuda").to(dtype).requires_grad_(True)
    g = torch.randn(4, M // 4, D, device="cuda").to(dtype)
    out = ops.gate_mix(pre, a, b)
    out.backward(g)
    got = (out.detach(), pre.grad, a.grad, b.grad)
    pre2, a2, b2 = (t.detach().clone().requires_grad_(True) for t in (pre, a, b))
    g1, g2 = torch.sigmoid(pre2).chunk(2, -1)
    ref_out = g1 * a2 + g2 * b2
    ref_out.backward(g)
    ref = (ref_out.detach(), pre2.grad, a2.grad, b2.grad)
    tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    for x, y in zip(got, ref):
        torch.testing.assert_close(x.float(), y.float(), **tol)
