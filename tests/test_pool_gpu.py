"""HGNetv2 stem pooling (max-pool 2x2 stride 1 of the zero-extended map) against
F.max_pool2d(F.pad(x, (0, 1, 0, 1)), 2, stride=1, ceil_mode=True) and its autograd."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C,H,W", [(2, 24, 20, 20), (1, 8, 5, 7), (3, 16, 33, 31), (2, 32, 160, 160)])
def test_stem_pool_matches_torch(hip_lib, dtype, B, C, H, W):
    from dfine_seg_amd import ops

    g = torch.Generator(device="cuda").manual_seed(B + C + H)
    # post-activation statistics: many exact zeros, some negatives (LAB), bf16 ties between neighbours
    x = torch.randn(B, C, H, W, device="cuda", generator=g)
    x = torch.where(x > 0.3, x, torch.where(x < -1.5, x, torch.zeros_like(x))).to(dtype)
    x = x.contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, C, H, W, device="cuda", generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(F.pad(xr, (0, 1, 0, 1)), 2, stride=1, ceil_mode=True)
    yr.backward(gy)
    xd = x.clone().requires_grad_(True)
    y = ops.stem_pool(xd)
    assert y.shape == yr.shape
    assert torch.equal(y, yr)
    y.backward(gy)
    # sums of up to 4 gradients: fp32 accumulation here, ATen accumulates in the map's dtype order-dependently
    tol = dict(rtol=0, atol=0) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(xd.grad.float(), xr.grad.float(), **tol)
