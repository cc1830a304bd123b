"""Fused FDR head (softmax -> Integral -> distance2bbox, LQE statistics) against the same chain in
plain PyTorch fp32 (the reference's formulation, dfine_decoder.py:276-313, arch/utils.py:119-142)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _reference(corners, ref, W, reg_scale, k):
    lead = corners.shape[:-1]
    prob = torch.softmax(corners.view(*lead, 4, -1), dim=-1)
    d = prob @ W
    rs = abs(reg_scale)
    wq, hq = ref[..., 2] / rs, ref[..., 3] / rs
    x1 = ref[..., 0] - (0.5 * rs + d[..., 0]) * wq
    y1 = ref[..., 1] - (0.5 * rs + d[..., 1]) * hq
    x2 = ref[..., 0] + (0.5 * rs + d[..., 2]) * wq
    y2 = ref[..., 1] + (0.5 * rs + d[..., 3]) * hq
    box = torch.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], -1)
    top = prob.topk(k, dim=-1).values
    stat = torch.cat([top, top.mean(-1, keepdim=True)], -1).view(*lead, -1)
    return box, stat


@pytest.mark.parametrize("B,L,n_bins", [(2, 7, 33), (32, 492, 33), (1, 1, 17), (3, 65, 33)])
def test_fdr_head_matches_torch(hip_lib, B, L, n_bins):
    from dfine_seg_amd import ops

    g = torch.Generator(device="cuda").manual_seed(B * 100 + L)
    corners = (torch.randn(B, L, 4 * n_bins, device="cuda", generator=g) * 2).requires_grad_(True)
    ref = torch.rand(B, L, 4, device="cuda", generator=g) * 0.5 + 0.1
    W = torch.linspace(-1.0, 1.0, n_bins, device="cuda") ** 3 * 2
    gb = torch.randn(B, L, 4, device="cuda", generator=g)
    gs = torch.randn(B, L, 20, device="cuda", generator=g)
    box_r, stat_r = _reference(corners, ref, W, 4.0, 4)
    (box_r * gb).sum().add((stat_r * gs).sum()).backward()
    want = corners.grad.clone()
    corners.grad = None
    box, stat = ops.fdr_head(corners, ref, W, 4.0, 4)
    torch.testing.assert_close(box, box_r, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(stat, stat_r, rtol=1e-5, atol=1e-6)
    (box * gb).sum().add((stat * gs).sum()).backward()
    torch.testing.assert_close(corners.grad, want, rtol=1e-4, atol=1e-6)
    # one consumer only: the other gradient arrives as None
    corners.grad = None
    box2, _ = ops.fdr_head(corners, ref, W, 4.0, 4)
    (box2 * gb).sum().backward()
    box_r2, _ = _reference(corners.detach().requires_grad_(True), ref, W, 4.0, 4)
    assert torch.isfinite(corners.grad).all()


def test_fdr_head_rejects_other_k(hip_lib):
    from dfine_seg_amd import _lib

    z = torch.zeros(4 * 33, device="cuda")
    assert hip_lib.dfine_fdr_head_fwd(_lib.ptr(z), _lib.ptr(z), _lib.ptr(z), _lib.ptr(z), _lib.ptr(z), _lib.ptr(z),
                                      1, 33, 3, 4.0, None) == -2
