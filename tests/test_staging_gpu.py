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
