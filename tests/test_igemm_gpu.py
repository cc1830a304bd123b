"""Implicit-GEMM kernel (csrc/igemm.hip) through the C ABI vs plain PyTorch fp32 convolution / linear on the
same bf16-rounded operands: forward and input gradient for every kind of dense conv on the path (HGNetv2
B0/B2/B5 + hybrid encoder shapes incl. D-FINE-n's 21-channel CSP branch, the 3-channel stem, stride 2, the
stem's 2x2 convs with bottom/right padding), channel-slice inputs / outputs (concat buffers), the BatchNorm
statistics epilogue, bias + ReLU, and the mask head's 2x2 stride-2 transposed conv.

Tolerance: operands are exact in bf16, accumulation is fp32, the result is rounded to bf16 once:
|err| <= 2^-8 |ref| + a few fp32 round-offs of the K-term sum -> rtol 1e-2, atol 1e-2 * rms(ref)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["auto", "register-staged", "lds-dma-ring"])
def igemm_body(request, monkeypatch):
    """Every case runs through the dispatcher's own choice and through each of the two kernel bodies forced for all
    launches (csrc/igemm.hip register-staged, csrc/igemm2.hip LDS-DMA ring; DFINE_IGEMM_V is read per call)."""
    if request.param != "auto":
        monkeypatch.setenv("DFINE_IGEMM_V", "1" if request.param == "register-staged" else "2")
    else:
        monkeypatch.delenv("DFINE_IGEMM_V", raising=False)
    return request.param


def _bf(t):
    return t.to(torch.bfloat16)


def _close(got, ref, what):
    scale = ref.float().pow(2).mean().sqrt().item() + 1e-6
    torch.testing.assert_close(got.float(), ref.float(), rtol=1e-2, atol=1e-2 * scale, msg=lambda m: f"{what}: {m}")


CONV_CASES = [
    # B, H, W, CI, CO, k, stride, pad, pad_br
    (2, 20, 20, 128, 128, 3, 1, 1, None),   # encoder RepVgg 3x3
    (2, 40, 40, 128, 128, 1, 1, 0, None),   # 1x1
    (1, 16, 24, 1280, 384, 1, 1, 0, None),  # wide aggregation
    (2, 33, 31, 48, 24, 3, 2, 1, None),     # stem3, stride 2, odd extent
    (2, 17, 19, 24, 12, 2, 1, 0, 1),        # stem2a: 2x2, pad bottom/right only
    (2, 17, 19, 12, 24, 2, 1, 0, 1),        # stem2b: CI = 12 (8-byte path)
    (2, 32, 32, 3, 24, 3, 2, 1, None),      # stem1: CI = 3 (element path)
    (2, 10, 10, 21, 21, 3, 1, 1, None),     # D-FINE-n c4 = 21
    (2, 10, 10, 298, 128, 1, 1, 0, None),   # D-FINE-n cv4 (K % 8 != 0)
    (2, 12, 12, 64, 21, 1, 1, 0, None),     # N = 21
    (3, 13, 11, 160, 48, 1, 1, 0, None),    # M not a multiple of the tile
    (1, 30, 30, 192, 192, 3, 1, 1, None),   # x encoder 3x3
    (1, 8, 8, 32, 32, 3, 1, 1, None),
    # more 3x3 stride-1 shapes: ragged right and bottom edges, N tails
    (2, 80, 80, 96, 64, 3, 1, 1, None),     # HGNetv2 stage 2 first layer
    (1, 160, 160, 32, 32, 3, 1, 1, None),   # HGNetv2 stage 1
    (2, 40, 40, 128, 128, 3, 1, 1, None),
    (3, 17, 23, 32, 48, 3, 1, 1, None),
    (2, 21, 37, 64, 136, 3, 1, 1, None),
    (1, 5, 3, 32, 32, 3, 1, 1, None),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_forward_and_input_grad(hip_lib, case):
    from dfine_seg_amd import gemm

    B, H, W, CI, CO, k, s, p, pbr = case
    g = torch.Generator(device="cuda").manual_seed(1)
    x = _bf(torch.randn(B, H, W, CI, device="cuda", generator=g))
    w = _bf(torch.randn(CO, k, k, CI, device="cuda", generator=g) / (k * k * CI) ** 0.5)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.float().permute(0, 3, 1, 2)
    xin = xr if pbr is None else F.pad(xr, (0, pbr, 0, pbr))
    ref = F.conv2d(xin, wr, None, s, p)
    stats = torch.zeros(8 * 2 * CO * 2, dtype=torch.float32, device="cuda")
    y = gemm.conv_forward(x, w, s, p, pbr, stats=stats)
    torch.cuda.synchronize()
    _close(y, ref.permute(0, 2, 3, 1), f"conv fwd {case}")
    # statistics epilogue: sums of the bf16-rounded outputs over all pixels
    sums = stats.view(torch.float64).view(8, 2, CO).sum(0)
    yf = y.float().view(-1, CO).double()
    torch.testing.assert_close(sums[0], yf.sum(0), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(sums[1], (yf * yf).sum(0), rtol=1e-4, atol=1e-3)
    # input gradient
    gy = _bf(torch.randn(ref.shape, device="cuda", generator=g)).permute(0, 2, 3, 1).contiguous()
    (gx_ref,) = torch.autograd.grad(ref, xr, gy.float().permute(0, 3, 1, 2))
    wt = w.permute(3, 1, 2, 0).contiguous()  # [CI, KH, KW, CO]
    gx = gemm.conv_input_grad(gy, wt, (H, W), s, p)  # bottom/right padding is implied by gy's extent
    torch.cuda.synchronize()
    _close(gx, gx_ref.permute(0, 2, 3, 1), f"conv dgrad {case}")


def test_conv_reads_and_writes_channel_slices(hip_lib):
    """Input = channels [16:80) of a 96-wide map, output = channels [32:96) of a 128-wide concat buffer."""
    from dfine_seg_amd import gemm

    g = torch.Generator(device="cuda").manual_seed(2)
    wide = _bf(torch.randn(2, 24, 24, 96, device="cuda", generator=g))
    x = wide[..., 16:80]
    w = _bf(torch.randn(64, 3, 3, 64, device="cuda", generator=g) / 24.0)
    cat = torch.full((2, 24, 24, 128), 7.0, dtype=torch.bfloat16, device="cuda")
    out = gemm.conv_forward(x, w, 1, 1, out=cat[..., 32:96])
    assert out.data_ptr() == cat[..., 32:96].data_ptr()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
    _close(cat[..., 32:96], ref, "slice conv")
    assert (cat[..., :32] == 7).all() and (cat[..., 96:] == 7).all()  # neighbours untouched
    # unaligned slice start (channel 4: 8-byte path) and odd width
    x2 = wide[..., 4:44]
    w2 = _bf(torch.randn(20, 1, 1, 40, device="cuda", generator=g) / 6.0)
    out2 = gemm.conv_forward(x2, w2, 1, 0, out=cat[..., 100:120])
    ref2 = F.conv2d(x2.float().permute(0, 3, 1, 2), w2.float().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    _close(out2, ref2, "unaligned slice conv")


@pytest.mark.parametrize("M,N,K,relu", [(15744, 256, 256, False), (600, 1024, 256, True), (9600, 80, 256, False), (300, 132, 256, False),
                                        (1000, 512, 4, False), (777, 192, 1024, True), (64, 2, 64, False)])
def test_linear_forward_bias_relu(hip_lib, M, N, K, relu):
    from dfine_seg_amd import gemm

    g = torch.Generator(device="cuda").manual_seed(3)
    x = _bf(torch.randn(M, K, device="cuda", generator=g))
    w = _bf(torch.randn(N, K, device="cuda", generator=g) / K ** 0.5)
    b = torch.randn(N, device="cuda", generator=g)
    y = gemm.linear_forward(x, w, b, relu)
    ref = F.linear(x.float(), w.float(), b)
    if relu:
        ref = ref.relu()
    _close(y, ref, f"linear {M}x{N}x{K}")


def test_conv_transpose_2x2(hip_lib):
    from dfine_seg_amd import gemm

    g = torch.Generator(device="cuda").manual_seed(4)
    B, H, W, CI, CO = 2, 9, 11, 64, 48
    x = _bf(torch.randn(B, H, W, CI, device="cuda", generator=g))
    w = _bf(torch.randn(CI, 2, 2, CO, device="cuda", generator=g) / 8.0)  # module weight [CI, CO, 2, 2] in channels_last memory
    wr = w.float().permute(0, 3, 1, 2)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.conv_transpose2d(xr, wr, None, 2)
    wt = w.permute(3, 1, 2, 0).contiguous()  # [CO, 2, 2, CI]
    stats = torch.zeros(8 * 2 * CO * 2, dtype=torch.float32, device="cuda")
    y = gemm.conv_transpose2x2_forward(x, wt, stats=stats)
    _close(y, ref.permute(0, 2, 3, 1), "convT fwd")
    sums = stats.view(torch.float64).view(8, 2, CO).sum(0)
    torch.testing.assert_close(sums[0], y.float().view(-1, CO).double().sum(0), rtol=1e-4, atol=1e-3)
    gy = _bf(torch.randn(B, 2 * H, 2 * W, CO, device="cuda", generator=g))
    (gx_ref,) = torch.autograd.grad(ref, xr, gy.float().permute(0, 3, 1, 2))
    gx = gemm.conv_transpose2x2_input_grad(gy, w)
    _close(gx, gx_ref.permute(0, 2, 3, 1), "convT dgrad")


def test_full_size_linearity_property(hip_lib):
    """BASELINE-size launch (m@640, batch 32: 3x3 128->128 on 80x80): conv(x1 + x2) == conv(x1) + conv(x2) on
    bf16-exact integer data, bit for bit, and a spot check of 64 random output pixels against fp32 conv2d."""
    from dfine_seg_amd import gemm

    g = torch.Generator(device="cuda").manual_seed(5)
    B, H, W, C = 32, 80, 80, 128
    x1 = torch.randint(-2, 3, (B, H, W, C), device="cuda", generator=g).to(torch.bfloat16)
    x2 = torch.randint(-2, 3, (B, H, W, C), device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randint(-2, 3, (C, 3, 3, C), device="cuda", generator=g).float() / 4).to(torch.bfloat16)
    ya, yb, yab = gemm.conv_forward(x1, w, 1, 1), gemm.conv_forward(x2, w, 1, 1), gemm.conv_forward(x1 + x2, w, 1, 1)
    # |sums| <= 9*128*4*0.5 = 2304 with a 1/4 grid: exactly representable in fp32, but not always in bf16 -> compare in fp32 within 1 bf16 ulp
    torch.testing.assert_close(yab.float(), ya.float() + yb.float(), rtol=2 ** -7, atol=0.25)
    ref = F.conv2d(x1[:2].float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), None, 1, 1).permute(0, 2, 3, 1)
    _close(ya[:2], ref, "full-size spot check")


def test_two_convs_of_one_input_form_one_input_gradient(hip_lib, monkeypatch):
    """ops.dense_conv_pair (RepVgg's 3x3 + 1x1, CSPLayer's two 1x1): the second input-gradient GEMM adds into the first's
    output in its epilogue (dfine_igemm_bf16 act | 0x400). The sum it forms -- bf16(bf16 dx_a + bf16 dx_b) -- is what autograd's
    accumulation computes from the two separate nodes: outputs and input gradient BIT-equal (weight gradients to the order of
    their float atomics), on RepVgg and on a CSP block of the encoder."""
    import copy

    from dfine_seg_amd.encoder import CSP, RepVgg
    from dfine_seg_amd.layers import ConvNorm

    torch.manual_seed(0)
    for make, x in ((lambda: RepVgg(64), torch.randn(3, 64, 20, 28, device="cuda")), (lambda: CSP(96, 64, 2), torch.randn(2, 96, 17, 23, device="cuda"))):
        mod = make().cuda().to(memory_format=torch.channels_last).train()
        ref = copy.deepcopy(mod)
        x = x.bfloat16().contiguous(memory_format=torch.channels_last)
        outs = []
        for m, paired in ((ref, False), (mod, True)):
            with monkeypatch.context() as mp:  # (not monkeypatch.undo(): the igemm_body fixture's environment must stay)
                if not paired:
                    mp.setattr(ConvNorm, "_pairable", lambda self, x: False)
                xx = x.clone().requires_grad_(True)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = m(xx)
                (y.float() * torch.linspace(-1, 1, y.numel(), device="cuda").view_as(y)).sum().backward()
                outs.append((y.detach(), xx.grad, {n: p.grad for n, p in m.named_parameters()}))
        (y0, g0, p0), (y1, g1, p1) = outs
        assert torch.equal(y0, y1) and torch.equal(g0, g1)
        for n in p0:  # weight gradients are sums of float atomics (order varies run to run), the rest are plain reductions
            torch.testing.assert_close(p0[n], p1[n], rtol=1e-4, atol=1e-4 * float(p0[n].abs().max()), msg=lambda m: f"{n}: {m}")


@pytest.mark.parametrize("B,H,W,CI,CO,k", [(3, 20, 20, 128, 128, 1), (2, 40, 40, 256, 64, 1), (4, 80, 80, 64, 128, 1), (2, 40, 40, 64, 96, 3),
                                           (2, 20, 20, 384, 128, 3), (1, 7, 9, 32, 48, 1), (8, 80, 80, 128, 128, 1)])
@pytest.mark.parametrize("act,lab,second", [(1, True, False), (2, False, True), (0, False, False)])
def test_input_grad_with_norm_backward_sums_in_the_epilogue(hip_lib, igemm_body, B, H, W, CI, CO, k, act, lab, second):
    """dfine_igemm_bf16_bnred: the input gradient is the plain launch's, and the four per-channel sums (over the 8 fp64
    replicas) are those of normact.hip's reduction pass computed here in fp64 torch from the same bf16 tensors:
    sum(dz), sum(dz * xhat), sum(d * act(z)), sum(d); dz = d * lab_scale * act'(z), d = dx + dy2 (1e-4 relative)."""
    from dfine_seg_amd import gemm

    if igemm_body != "auto":
        pytest.skip("the bnred launch picks its body by the operand layout alone")
    torch.manual_seed(5)
    dev = "cuda"
    gy = _bf(torch.randn(B, H, W, CO, device=dev))
    wt = _bf(torch.randn(CI, k, k, CO, device=dev) / (CO * k * k) ** 0.5)
    x = _bf(torch.randn(B, H, W, CI, device=dev) * 1.3 + 0.2)
    wide = _bf(torch.randn(B, H, W, CI + 24, device=dev))
    dy2 = wide[..., 8 : 8 + CI] if second else None
    mean, rstd = torch.randn(CI, device=dev) * 0.1 + 0.2, torch.rand(CI, device=dev) + 0.5
    gamma, beta = torch.rand(CI, device=dev) + 0.5, torch.randn(CI, device=dev) * 0.3
    ls = torch.tensor([1.7], device=dev) if lab else None
    red = torch.zeros(64 * CI, device=dev)
    plain = gemm.conv_input_grad(gy, wt, (H, W), 1, k // 2)
    dx = gemm.conv_input_grad(gy, wt, (H, W), 1, k // 2, bnred=gemm.BnRed(x, dy2, mean, rstd, gamma, beta, ls, red, act))
    assert dx._bnred_done
    assert torch.equal(dx, plain)
    d = dx.double() + (dy2.double() if second else 0)
    xh = (x.double() - mean.double()) * rstd.double()
    z = xh * gamma.double() + beta.double()
    if act == 1:
        g, f = (z > 0).double(), z.clamp_min(0)
    elif act == 2:
        s = torch.sigmoid(z)
        g, f = s * (1 + z * (1 - s)), z * s
    else:
        g, f = torch.ones_like(z), z
    dz = d * (1.7 if lab else 1.0) * g
    want = [dz.sum((0, 1, 2)), (dz * xh).sum((0, 1, 2))] + ([(d * f).sum((0, 1, 2)), d.sum((0, 1, 2))] if lab else [])
    got = red.view(torch.float64).view(8, 4, CI).sum(0)
    for q, w in enumerate(want):
        rel = float((got[q] - w).norm() / w.norm())
        assert rel < 1e-4, (q, rel)
    if not lab:
        assert float(got[2:].abs().max()) == 0.0
