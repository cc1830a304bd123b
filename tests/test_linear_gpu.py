"""Split-M MFMA weight/bias-gradient kernel of Linear layers and dense convolutions against a plain
PyTorch fp32 reference of the same contraction (dW = dY^T X, db = sum dY; conv: autograd of F.conv2d --
on the bf16-rounded operands, fp32 math)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(15744, 1024, 256), (15744, 256, 1024), (15744, 80, 256), (15744, 192, 256),
                                   (268800, 256, 256), (12800, 256, 256), (1000, 8, 8), (33, 72, 136), (1, 64, 64),
                                   # every tile shape: 128x128, 128x64, 64x128 (and the fallback to smaller tiles)
                                   (40000, 128, 128), (40000, 384, 64), (40000, 64, 384), (300000, 128, 256),
                                   (2000, 256, 256), (700, 136, 264),
                                   # N or K not a multiple of 8 (bbox heads 4 / 132 outputs, LQE's 20 inputs)
                                   (15744, 132, 256), (15744, 4, 256), (15744, 64, 20), (5000, 2, 6), (999, 70, 30)])
def test_linear_wgrad_matches_fp32_reference(hip_lib, M, N, K):
    from dfine_seg_amd import _lib

    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    dy = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    dw = torch.zeros(N, K, device="cuda")
    db = torch.zeros(N, device="cuda")
    rc = hip_lib.dfine_linear_wgrad(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw), _lib.ptr(db), M, N, K, K, N, K, _lib.stream_ptr())
    assert rc == 0
    ref_w = dy.float().t() @ x.float()
    ref_b = dy.float().sum(0)
    # exact bf16 products, fp32 sums in a different order (MFMA blocks + atomics): error ~ sqrt(M) * eps
    tol = 3e-8 * M + 2e-4  # fp32 round-off of a length-M sum of O(1) terms
    torch.testing.assert_close(dw, ref_w, rtol=1e-4, atol=tol * 4)
    torch.testing.assert_close(db, ref_b, rtol=1e-4, atol=tol * 4)
    # accumulation semantics: a second call adds on top
    hip_lib.dfine_linear_wgrad(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw), None, M, N, K, K, N, K, _lib.stream_ptr())
    torch.testing.assert_close(dw, 2 * ref_w, rtol=1e-4, atol=tol * 8)


def test_unsupported_shape_is_rejected(hip_lib):
    from dfine_seg_amd import _lib

    t = torch.zeros(16, 13, device="cuda", dtype=torch.bfloat16)
    out = torch.zeros(13, 13, device="cuda")
    assert hip_lib.dfine_linear_wgrad(_lib.ptr(t), _lib.ptr(t), _lib.ptr(out), None, 16, 13, 13, 12, 13, 13, None) == -1  # ldw < K
    # odd N / K (rows only 2-byte aligned) are supported: element-wise loads
    x = torch.randn(16, 13, device="cuda").bfloat16()
    assert hip_lib.dfine_linear_wgrad(_lib.ptr(x), _lib.ptr(x), _lib.ptr(out), None, 16, 13, 13, 13, 13, 13, None) == 0
    torch.testing.assert_close(out, x.float().t() @ x.float(), rtol=1e-4, atol=1e-4)


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


@pytest.mark.parametrize("B,H,W,CI,CO,k,stride,pad", [
    (2, 40, 40, 32, 32, 3, 1, 1),    # HG-block 3x3
    (3, 23, 17, 64, 48, 3, 1, 1),    # ragged extent, borders everywhere
    (2, 40, 36, 32, 64, 3, 2, 1),    # stem3 style stride 2
    (2, 33, 33, 16, 16, 2, 1, 0),    # stem2a/2b style 2x2 without padding
    (2, 20, 20, 128, 128, 3, 1, 1),  # RepVgg 3x3: K = 1152, 128-wide tiles
    (4, 80, 80, 64, 64, 3, 1, 1),    # many pixels: several m-chunks
    (2, 20, 20, 256, 64, 1, 1, 0),   # 1x1 through the conv entry
    (1, 9, 9, 8, 8, 5, 1, 2),        # 5x5, tiny
    (2, 16, 16, 24, 40, 3, 1, 1),    # channel counts that are multiples of 8 only
    (2, 33, 33, 24, 12, 2, 1, 0),    # stem2a: 12 output channels
    (2, 33, 33, 12, 24, 2, 1, 0),    # stem2b: 12 input channels (chunks straddle taps)
    (2, 32, 32, 3, 24, 3, 2, 1),     # stem1: the 3-channel image
    (2, 10, 10, 21, 21, 3, 1, 1),    # D-FINE-n's 21-channel CSP branch
])
def test_conv_wgrad_matches_fp32_reference(hip_lib, B, H, W, CI, CO, k, stride, pad):
    import torch.nn.functional as F

    from dfine_seg_amd import _lib

    g = torch.Generator(device="cuda").manual_seed(B * H + CI + CO + k)
    x = torch.randn(B, CI, H, W, device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    gy = torch.randn(B, CO, OH, OW, device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.zeros(CO, CI, k, k, device="cuda", requires_grad=True)
    F.conv2d(x.float(), w, None, stride, pad).backward(gy.float())
    dw = torch.zeros(CO, k, k, CI, device="cuda")  # [K][KH][KW][C]
    rc = hip_lib.dfine_conv_wgrad(_lib.ptr(gy.permute(0, 2, 3, 1)), _lib.ptr(x.permute(0, 2, 3, 1)), _lib.ptr(dw),
                                  B, H, W, CI, OH, OW, CO, k, k, stride, pad, CO, CI, _lib.stream_ptr())
    assert rc == 0
    M = B * OH * OW
    tol = (3e-8 * M + 2e-4) * 4
    torch.testing.assert_close(dw.permute(0, 3, 1, 2), w.grad, rtol=1e-4, atol=tol)


def test_conv_wgrad_rejects_inconsistent_geometry(hip_lib):
    from dfine_seg_amd import _lib

    t = torch.zeros(1, 8, 8, 8, device="cuda", dtype=torch.bfloat16)
    out = torch.zeros(8, 3, 3, 8, device="cuda")
    # 8x8 input, 3x3 pad 1 stride 1: an 11x11 output would start its last taps beyond the map
    assert hip_lib.dfine_conv_wgrad(_lib.ptr(t), _lib.ptr(t), _lib.ptr(out), 1, 8, 8, 8, 11, 11, 8, 3, 3, 1, 1, 8, 8, None) == -1
    assert hip_lib.dfine_conv_wgrad(_lib.ptr(t), _lib.ptr(t), _lib.ptr(out), 1, 8, 8, 8, 8, 8, 8, 3, 3, 1, 1, 4, 8, None) == -1  # pitch < channels


@pytest.mark.parametrize("k,stride,pad", [(3, 1, 1), (3, 2, 1), (2, 1, 0)])
def test_dense_conv_matches_torch(hip_lib, k, stride, pad):
    """kxk conv through ops.dense_conv (bf16 mode) vs F.conv2d autograd in fp32 on the same values, with
    the gradient accumulated into a channels_last fp32 .grad and returned through autograd."""
    import torch.nn.functional as F

    from dfine_seg_amd import ops

    g = torch.Generator(device="cuda").manual_seed(11 + k)
    x = torch.randn(2, 32, 22, 26, device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(48, 32, k, k, device="cuda", generator=g) * 0.1).contiguous(memory_format=torch.channels_last)
    xr, wr = x.float().requires_grad_(True), w.bfloat16().float().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, pad)
    gy = torch.randn(yr.shape, device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    yr.backward(gy.float())
    for direct in (True, False):
        xd, wd = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        if direct:
            wd.grad = torch.zeros_like(wd)
            assert wd.grad.is_contiguous(memory_format=torch.channels_last)
        y = ops.dense_conv(xd, wd, stride, pad)
        y.backward(gy)
        torch.testing.assert_close(y.float(), yr.detach(), rtol=2e-2, atol=3e-2)
        torch.testing.assert_close(xd.grad.float(), xr.grad, rtol=2e-2, atol=3e-2)
        torch.testing.assert_close(wd.grad, wr.grad, rtol=1e-3, atol=3e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [(256, 256, 256, 4), (4, 512, 256, 1), (256, 64, 1, 20), (256, 1024, 256, 33)])
def test_mlp_chain_relu_backward_in_the_next_layers_epilogue(hip_lib, dims):
    """layers.MLP (reference dfine_decoder.py:33-46): every hidden ReLU's backward is applied by the next layer's input-gradient
    GEMM (dfine_igemm_bf16_relumask; shapes off the 16-byte paths take the GEMM + threshold_backward route) -- against the same
    stack through torch.nn.functional in fp32 on the bf16-rounded weights: outputs 2e-2, input gradient 5e-2 relative L2 (two
    bf16-rounded hidden activations and three bf16-rounded gradients in the chain); and
    bit-equal to the separate-mask formulation (relu_in / mask_by_consumer off)."""
    from dfine_seg_amd import ops
    from dfine_seg_amd.layers import MLP

    d_in, d_hidden, d_out, tail = dims
    torch.manual_seed(d_in + d_hidden)
    mlp = MLP(d_in, d_hidden, d_out, 3).cuda()
    x = torch.randn(7, 300 + tail, d_in, device="cuda").bfloat16()
    r = torch.randn(7, 300 + tail, d_out, device="cuda")

    def run(fused):
        xx = x.clone().requires_grad_()
        for p in mlp.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            if fused:
                y = mlp(xx)
            else:
                y = xx
                for i, layer in enumerate(mlp.layers):
                    y = ops.linear(y, layer.weight, layer.bias, relu=i < 2)
        (y.float() * r).sum().backward()
        return y.detach().float(), xx.grad.float(), [p.grad.float().clone() for p in mlp.parameters()]

    y1, gx1, gp1 = run(True)
    y0, gx0, gp0 = run(False)
    assert torch.equal(y1, y0) and torch.equal(gx1, gx0)
    for a, b in zip(gp1, gp0):  # weight gradients: float atomics, order varies
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()) + 1e-12)
    xr = x.float().requires_grad_()
    h = xr
    for i, layer in enumerate(mlp.layers):
        h = torch.nn.functional.linear(h, layer.weight.detach().bfloat16().float(), layer.bias.detach().float())
        h = torch.relu(h) if i < 2 else h
    (h * r).sum().backward()
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-12))
    assert rel(y1, h.detach()) < 2e-2 and rel(gx1, xr.grad) < 5e-2
