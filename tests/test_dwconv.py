"""Depthwise conv HIP kernels (fwd / bwd-data / bwd-weight) against a plain PyTorch fp32
reference of the same op (F.conv2d with groups=C, computed on the CPU in fp32)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [  # (B, C, H, W, k, stride): HGNetv2-B2/B0 light blocks + downsamples, SCDown, ragged sizes
    (2, 128, 40, 40, 5, 1), (2, 256, 20, 20, 5, 1), (2, 96, 160, 160, 3, 2), (2, 384, 80, 80, 3, 2),
    (2, 768, 40, 40, 3, 2), (2, 256, 80, 80, 3, 2), (3, 64, 21, 37, 5, 1), (1, 32, 7, 5, 3, 1),
    (2, 512, 15, 15, 5, 2), (1, 8, 1, 1, 3, 1),
]


@pytest.mark.parametrize("B,C,H,W,k,stride", SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_depthwise_conv_matches_torch_fp32(hip_lib, B, C, H, W, k, stride, dtype):
    from dfine_seg_amd import ops

    g = torch.Generator().manual_seed(C + H + k)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, 1, k, k, generator=g) / k
    xq = x.to(dtype).float()  # the reference sees the same (rounded) input values
    x_ref = xq.clone().requires_grad_(True)
    w_ref = w.clone().requires_grad_(True)
    y_ref = F.conv2d(x_ref, w_ref, None, stride, (k - 1) // 2, 1, C)
    gy = torch.randn(y_ref.shape, generator=g).to(dtype).float()
    y_ref.backward(gy)

    xd = x.to(dtype).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = w.cuda().requires_grad_(True)
    y = ops.depthwise_conv2d(xd, wd, stride)
    assert y.shape == y_ref.shape and y.dtype == dtype
    y.backward(gy.to(dtype).cuda())
    # fp32: accumulation-order noise only. bf16: + one rounding of y / dx to 8 mantissa bits.
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=1e-2, atol=2e-2)
    torch.testing.assert_close(y.float().cpu(), y_ref.detach(), **tol)
    torch.testing.assert_close(xd.grad.float().cpu(), x_ref.grad, **tol)
    # dw sums B*OH*OW products in fp32 (atomics): scale the tolerance with the reduction length
    n = B * y_ref.shape[2] * y_ref.shape[3]
    torch.testing.assert_close(wd.grad.cpu(), w_ref.grad, rtol=1e-3, atol=1e-4 * n**0.5 + 1e-4)


def test_unsupported_shapes_fail_loudly(hip_lib):
    from dfine_seg_amd import _lib, ops

    x = torch.randn(1, 12, 8, 8, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(12, 1, 3, 3, device="cuda")
    with pytest.raises(_lib.DfineHipError):
        ops.depthwise_conv2d(x, w, 1)  # C % 8 != 0 for bf16


@pytest.mark.parametrize("B,C,H,W,k", [(2, 128, 40, 40, 5), (3, 64, 21, 37, 5), (1, 32, 7, 5, 3), (2, 256, 20, 20, 5)])
def test_statistics_epilogue_of_the_strip_kernel(hip_lib, B, C, H, W, k):
    """dfine_dwconv_fwd with `stats`: per channel the sum and sum of squares of the bf16-ROUNDED outputs, spread over 8 fp64
    replicas -- exactly what a statistics pass over y would give (1e-6 relative: fp32 partials inside a workgroup)."""
    from dfine_seg_amd import ops

    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, C, H, W, generator=g).bfloat16().cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, 1, k, k, generator=g) / k).cuda()
    ws = torch.zeros(34 * C, device="cuda")
    with torch.no_grad():
        y = ops.depthwise_conv2d(x, w, 1, stats=ws[: 32 * C])
        y_plain = ops.depthwise_conv2d(x, w, 1)
    assert torch.equal(y, y_plain)
    sums = ws[: 32 * C].view(torch.float64).view(8, 2, C).sum(0)
    yd = y.double()
    torch.testing.assert_close(sums[0], yd.sum((0, 2, 3)), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(sums[1], (yd * yd).sum((0, 2, 3)), rtol=1e-5, atol=1e-3)
