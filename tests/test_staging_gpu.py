"""Input staging of the GT masks (targets.MaskStager + dfine_mask_targets_nearest): bit-exact against the reference's
per-image formulation (dfine_criterion.py:239-270: m.float() -> F.interpolate(size, mode="nearest") -> clamp(0, 1)), for the
exact 4:1 case of the model (640 -> 160) and for non-integer ratios, ragged GT counts, images without masks; the pinned
double buffer survives being reused while earlier transfers are still in flight."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(masks, G, out_h, out_w):
    out = torch.zeros(len(masks), G, out_h, out_w)
    for i, m in enumerate(masks):
        if m is None or m.numel() == 0:
            continue
        out[i, : m.shape[0]] = F.interpolate(m[None].float(), size=(out_h, out_w), mode="nearest")[0].clamp_(0, 1)
    return out


@pytest.mark.parametrize("H,W,oh,ow", [(640, 640, 160, 160), (96, 130, 37, 51), (64, 64, 64, 64), (50, 70, 100, 35)])
def test_mask_targets_are_bit_exact(hip_lib, H, W, oh, ow):
    from dfine_seg_amd.targets import MaskStager

    g = torch.Generator().manual_seed(H + ow)
    masks = [(torch.rand(n, H, W, generator=g) > 0.6).to(torch.uint8) * (3 if n == 2 else 1) if n else None for n in (3, 0, 5, 2)]
    out = MaskStager.for_device("cuda").stage(masks, 5, oh, ow)
    assert out is not None and out.dtype == torch.float32
    assert torch.equal(out.cpu(), _ref(masks, 5, oh, ow))


def test_pinned_double_buffer_is_not_overwritten_in_flight(hip_lib):
    from dfine_seg_amd.targets import MaskStager

    stager = MaskStager.for_device("cuda")
    g = torch.Generator().manual_seed(0)
    batches = [[(torch.rand(4, 320, 320, generator=g) > 0.5).to(torch.uint8) for _ in range(8)] for _ in range(6)]
    torch.cuda.synchronize()
    torch.cuda._sleep(200_000_000)  # keep the GPU busy so all six stagings are enqueued before the first copy runs
    outs = [stager.stage(b, 4, 80, 80) for b in batches]
    torch.cuda.synchronize()
    for b, o in zip(batches, outs):
        assert torch.equal(o.cpu(), _ref(b, 4, 80, 80))


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,size", [(64, 96, None), (64, 96, (128, 160)), (64, 96, (32, 64)), (37, 53, (96, 96)), (640, 640, (576, 576))])
def test_image_staging_matches_normalize_then_interpolate(hip_lib, H, W, size):
    """ImageStager vs the reference's host pipeline restated with plain PyTorch on the CPU: albumentations' A.Normalize
    arithmetic ((u8 - mean*255) * float32(1 / (std*255))), ToTensorV2, stack, and the collate function's
    F.interpolate(bilinear, align_corners=False). Unresized: bit-exact. Resized: within 2e-6 (values are O(1); the CPU
    kernel and this one may contract the 4-tap sum differently). bf16 output = the fp32 result rounded once."""
    from dfine_seg_amd.targets import ImageStager

    g = torch.Generator().manual_seed(H + W)
    imgs = [torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8) for _ in range(3)]
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    sub = torch.tensor(mean, dtype=torch.float32) * 255.0
    mul = torch.reciprocal(torch.tensor(std, dtype=torch.float32) * 255.0)
    ref = torch.stack([((im.float() - sub) * mul).permute(2, 0, 1) for im in imgs])
    if size is not None:
        ref = torch.nn.functional.interpolate(ref, size=size, mode="bilinear", align_corners=False)
    st = ImageStager("cuda")
    out = st.stage(imgs, size=size, mean=mean, std=std)
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    if size is None:
        assert torch.equal(out.cpu(), ref)
    else:
        assert float((out.cpu() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    out16 = st.stage(imgs, size=size, mean=mean, std=std, dtype=torch.bfloat16)
    assert float((out16.float().cpu() - ref).abs().max()) <= 2 ** -8 * float(ref.abs().max()) + 2e-6
    # the reference's default normalisation (mean 0, std 1): pixel / 255 as a reciprocal multiply
    out01 = st.stage(torch.stack(imgs))
    assert torch.equal(out01.cpu(), torch.stack([(im.float() * torch.reciprocal(torch.tensor(255.0))).permute(2, 0, 1) for im in imgs]))


@pytest.mark.gpu
def test_mask_targets_through_a_multiscale_resize(hip_lib):
    """via=(Hv, Wv): the masks were NOT resized on the host; the kernel composes the collate function's nearest resize
    (dataset.py:663-668) with the criterion's (dfine_criterion.py:239-270). Bit-exact."""
    from dfine_seg_amd.targets import MaskStager

    g = torch.Generator().manual_seed(3)
    H, W, Hv, Wv, G = 64, 96, 128, 160, 5
    masks = [(torch.rand(n, H, W, generator=g) > 0.5).to(torch.uint8) for n in (3, 0, 5)]
    oh, ow = Hv // 4, Wv // 4
    st = MaskStager("cuda")
    got = st.stage(masks, G, oh, ow, via=(Hv, Wv)).cpu()
    for i, m in enumerate(masks):
        if m.shape[0] == 0:
            assert float(got[i].abs().max()) == 0
            continue
        mid = (torch.nn.functional.interpolate(m.unsqueeze(1).float(), size=(Hv, Wv), mode="nearest").squeeze(1) > 0.5).to(torch.uint8)
        ref = torch.nn.functional.interpolate(mid.unsqueeze(1).float(), size=(oh, ow), mode="nearest").squeeze(1).clamp(0, 1)
        assert torch.equal(got[i, : m.shape[0]], ref)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [((10, 12), (20, 24)), ((5, 6), (20, 24)), ((7, 9), (20, 25)), ((20, 24), (20, 24))])
def test_bilinear_add_matches_interpolate(dtype, shape):
    """``base + F.interpolate(x, size, mode="bilinear", align_corners=False)`` (MaskPixelDecoder's top-down sum) forward, and
    the resize's backward as a gather against ATen's scatter, on x2, x4, a non-integer ratio and the identity. fp32: 1e-6
    (same arithmetic); bf16: the fused form rounds once where ATen rounds the resized map and the sum (one bf16 ulp)."""
    import torch.nn.functional as F

    from dfine_seg_amd import ops

    torch.manual_seed(2)
    (h, w), (H, W) = shape
    B, C = 3, 32
    x = torch.randn(B, C, h, w, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    base = torch.randn(B, C, H, W, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    g = torch.randn(B, C, H, W, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    res = []
    for fused in (True, False):
        xs, bs = x.clone().requires_grad_(True), base.clone().requires_grad_(True)
        if fused:
            y = ops.bilinear_add(bs, xs)
        else:
            y = bs.float() + F.interpolate(xs.float(), size=(H, W), mode="bilinear", align_corners=False)
        y.backward(g.to(y.dtype))
        res.append((y.detach().float(), xs.grad.float(), bs.grad.float()))
    (y1, gx1, gb1), (y0, gx0, gb0) = res
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=8e-3, atol=2e-2)
    torch.testing.assert_close(y1, y0, **tol)
    torch.testing.assert_close(gx1, gx0, **tol)
    assert torch.equal(gb1, gb0)
