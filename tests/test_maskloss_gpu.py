"""Fused mask-loss row statistics (focal-BCE sum, sum p*t, sum p) and their backward against the plain
PyTorch formulation of the reference's _focal_loss_mask / _dice_loss (dfine_criterion.py:272-312)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _reference(z, t, alpha, groups):
    B, M, HW = z.shape
    G = t.shape[1]
    z5 = z.float().view(B, groups, G, HW)
    t5 = t.unsqueeze(1).expand_as(z5)
    a5 = alpha.view(B, 1, G, 1)
    p = torch.sigmoid(z5)
    bce = F.binary_cross_entropy_with_logits(z5, t5, reduction="none")
    p_t = p * t5 + (1 - p) * (1 - t5)
    focal = ((a5 * t5 + (1 - a5) * (1 - t5)) * (1 - p_t) ** 2 * bce).sum(-1)
    return torch.stack([focal, (p * t5).sum(-1), p.sum(-1)], -1).view(B, M, 3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,G,groups,HW", [(2, 3, 1, 1600), (2, 4, 5, 25600), (1, 1, 2, 64), (3, 8, 12, 6400)])
def test_mask_row_stats_match_torch(hip_lib, dtype, B, G, groups, HW):
    from dfine_seg_amd import ops

    gen = torch.Generator(device="cuda").manual_seed(B * 7 + G + HW)
    M = groups * G
    z = (torch.randn(B, M, HW, device="cuda", generator=gen) * 4).to(dtype)
    t = (torch.rand(B, G, HW, device="cuda", generator=gen) > 0.7).float()
    t[0, 0, : HW // 8] = 0.5  # a non-binary stretch: the kernel does not assume {0, 1}
    alpha = 0.25 + 0.5 * torch.rand(B, G, device="cuda", generator=gen)
    gs = torch.randn(B, M, 3, device="cuda", generator=gen)
    zr = z.detach().float().clone().requires_grad_(True)
    ref = _reference(zr, t, alpha, groups)
    (ref * gs).sum().backward()
    zd = z.detach().clone().requires_grad_(True)
    stats = ops.mask_row_stats(zd, t, alpha)
    torch.testing.assert_close(stats, ref.detach(), rtol=1e-4, atol=2e-3)  # fp32 sums of HW terms in another order, fast exp/log
    (stats * gs).sum().backward()
    assert zd.grad.dtype == dtype
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=2e-2)
    torch.testing.assert_close(zd.grad.float(), zr.grad, **tol)


def test_mask_loss_rejects_bad_shapes(hip_lib):
    from dfine_seg_amd import _lib

    z = torch.zeros(2 * 3 * 12, device="cuda")
    assert hip_lib.dfine_mask_loss_stats(_lib.ptr(z), 0, _lib.ptr(z), _lib.ptr(z), _lib.ptr(z), 2, 3, 2, 12, None) == -1  # M % G
    assert hip_lib.dfine_mask_loss_stats(_lib.ptr(z), 0, _lib.ptr(z), _lib.ptr(z), _lib.ptr(z), 2, 3, 3, 10, None) == -2  # HW % 4


@pytest.mark.parametrize("B,G,groups,H,W", [(2, 4, 5, 16, 24), (3, 8, 13, 40, 32), (1, 3, 2, 8, 16), (2, 5, 54, 16, 16)])
def test_fused_mask_rows_loss_matches_the_materialised_path(hip_lib, B, G, groups, H, W):
    """dfine_mask_rows_{stats,dz,dfeat} (GEMM + loss statistics in one kernel each way, logits only in accumulators) against
    the path that materialises them: ``mask_row_stats(mask_logits_rows(rows, feat))`` (reference dfine_decoder.py:937-940 +
    dfine_criterion.py:272-357). Same bf16-rounded logits in both, so the statistics agree to fp32 summation order (and the
    fast exp/log); gradients to bf16 rounding of dz. Row counts that are not multiples of 8 or of the 128-row tile included."""
    from dfine_seg_amd import ops

    gen = torch.Generator(device="cuda").manual_seed(B * 31 + G + H)
    C, N, HW = 256, groups * G, H * W
    feat = (torch.randn(B, C, H, W, device="cuda", generator=gen) * 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    rows = (torch.randn(B, N, C, device="cuda", generator=gen) * 0.2).to(torch.bfloat16)
    t = (torch.rand(B, G, HW, device="cuda", generator=gen) > 0.7).float()
    t[0, 0, : HW // 8] = 0.5
    alpha = 0.25 + 0.5 * torch.rand(B, G, device="cuda", generator=gen)
    gs = torch.randn(B, N, 3, device="cuda", generator=gen)
    assert ops.mask_rows_loss_supported(rows, feat)

    r0, f0 = rows.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    ref = ops.mask_row_stats(ops.mask_logits_rows(r0, f0).reshape(B, N, HW), t, alpha)
    (ref * gs).sum().backward()
    r1, f1 = rows.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    got = ops.mask_rows_loss(r1, f1, t, alpha)
    torch.testing.assert_close(got, ref.detach(), rtol=2e-4, atol=2e-3 * HW ** 0.5)
    (got * gs).sum().backward()
    torch.cuda.synchronize()

    def rel(a, b):
        return float((a.float() - b.float()).norm() / b.float().norm().clamp(min=1e-12))

    # the materialised path rounds dz to bf16 too, then runs library GEMMs with fp32 accumulation: same roundings, other order
    assert rel(r1.grad, r0.grad) < 1e-2, rel(r1.grad, r0.grad)
    assert rel(f1.grad, f0.grad) < 1e-2, rel(f1.grad, f0.grad)
    assert f1.grad.shape == feat.shape and f1.grad.permute(0, 2, 3, 1).is_contiguous()
