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
