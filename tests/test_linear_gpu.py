"""Split-M MFMA weight/bias-gradient kernel of Linear layers against a plain PyTorch fp32 reference
of the same contraction (dW = dY^T X, db = sum dY on the bf16-rounded operands, fp32 math)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(15744, 1024, 256), (15744, 256, 1024), (15744, 80, 256), (15744, 192, 256),
                                   (268800, 256, 256), (12800, 256, 256), (1000, 8, 8), (33, 72, 136), (1, 64, 64)])
def test_linear_wgrad_matches_fp32_reference(hip_lib, M, N, K):
    from dfine_seg_amd import _lib

    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    dy = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    dw = torch.zeros(N, K, device="cuda")
    db = torch.zeros(N, device="cuda")
    rc = hip_lib.dfine_linear_wgrad(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw), _lib.ptr(db), M, N, K, K, _lib.stream_ptr())
    assert rc == 0
    ref_w = dy.float().t() @ x.float()
    ref_b = dy.float().sum(0)
    # exact bf16 products, fp32 sums in a different order (MFMA blocks + atomics): error ~ sqrt(M) * eps
    tol = 3e-8 * M + 2e-4  # fp32 round-off of a length-M sum of O(1) terms
    torch.testing.assert_close(dw, ref_w, rtol=1e-4, atol=tol * 4)
    torch.testing.assert_close(db, ref_b, rtol=1e-4, atol=tol * 4)
    # accumulation semantics: a second call adds on top
    hip_lib.dfine_linear_wgrad(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw), None, M, N, K, K, _lib.stream_ptr())
    torch.testing.assert_close(dw, 2 * ref_w, rtol=1e-4, atol=tol * 8)


def test_unsupported_shape_is_rejected(hip_lib):
    from dfine_seg_amd import _lib

    t = torch.zeros(16, 12, device="cuda", dtype=torch.bfloat16)
    out = torch.zeros(12, 12, device="cuda")
    assert hip_lib.dfine_linear_wgrad(_lib.ptr(t), _lib.ptr(t), _lib.ptr(out), None, 16, 12, 12, 12, None) == -2


def test_ops_linear_matches_torch_autograd(hip_lib):
    """ops.linear in bf16 mode (direct accumulation into pre-existing fp32 .grad, and the fallback
    that returns gradients) vs F.linear autograd in fp32 on the same bf16-rounded values."""
    import torch.nn.functional as F

    from dfine_seg_amd import ops

    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(4, 123, 256, device="cuda", generator=g).bfloat16()
    w = (torch.randn(768, 256, device="cuda", generator=g) * 0.05)
    b = torch.randn(768, device="cuda", generator=g) * 0.1
    gy = torch.randn(4, 123, 512, device="cuda", generator=g).bfloat16()
    xr = x.float().requires_grad_(True)
    wr, br = w.bfloat16().float().requires_grad_(True), b.bfloat16().float().requires_grad_(True)
    F.linear(xr, wr[:512], br[:512]).backward(gy.float())
    for direct in (True, False):
        xd = x.clone().requires_grad_(True)
        wd, bd = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        if direct:
            wd.grad, bd.grad = torch.zeros_like(wd), torch.zeros_like(bd)
        y = ops.linear(xd, wd, bd, 0, 512)
        assert y.dtype == torch.bfloat16
        y.backward(gy)
        torch.testing.assert_close(xd.grad.float(), xr.grad, rtol=2e-2, atol=2e-2)
        torch.testing.assert_close(wd.grad, wr.grad, rtol=1e-3, atol=2e-3)
        torch.testing.assert_close(bd.grad, br.grad, rtol=1e-3, atol=2e-3)
        assert float(wd.grad[512:].abs().max()) == 0.0


def test_pointwise_conv_matches_torch(hip_lib):
    """1x1 conv through ops.pointwise_conv (bf16 mode) vs F.conv2d autograd in fp32 on the same values."""
    import torch.nn.functional as F

    from dfine_seg_amd import ops

    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn(2, 96, 20, 24, device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(48, 96, 1, 1, device="cuda", generator=g) * 0.1
    gy = torch.randn(2, 48, 20, 24, device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    xr, wr = x.float().requires_grad_(True), w.bfloat16().float().requires_grad_(True)
    F.conv2d(xr, wr).backward(gy.float())
    for direct in (True, False):
        xd, wd = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        if direct:
            wd.grad = torch.zeros_like(wd)
        y = ops.pointwise_conv(xd, wd)
        y.backward(gy)
        torch.testing.assert_close(y.float(), F.conv2d(xr, wr).detach(), rtol=2e-2, atol=2e-2)
        torch.testing.assert_close(xd.grad.float(), xr.grad, rtol=2e-2, atol=2e-2)
        torch.testing.assert_close(wd.grad, wr.grad, rtol=1e-3, atol=3e-3)
