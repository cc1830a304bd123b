"""Host side of the GT-mask staging (targets.MaskStager): the pixels the loss will read are picked on the host with PyTorch's
nearest source-index rule, so only they cross PCIe. CPU test of that index against F.interpolate(mode="nearest"), alone and
composed with the multiscale augmentation's first resize (reference dfine_criterion.py:239-270, src/dl/dataset.py:645-670)."""
import pytest
import torch
import torch.nn.functional as F

from dfine_seg_amd.targets import nearest_source_index


@pytest.mark.parametrize("H,W,hv,wv,oh,ow", [(640, 640, 0, 0, 160, 160), (480, 640, 576, 704, 144, 176), (333, 517, 0, 0, 80, 129),
                                             (640, 640, 512, 512, 128, 128), (97, 61, 0, 0, 97, 61), (50, 70, 200, 210, 25, 30)])
def test_selection_equals_nearest_interpolation(H, W, hv, wv, oh, ow):
    torch.manual_seed(0)
    m = (torch.rand(3, H, W) > 0.5).to(torch.uint8)
    ref = m[None].float()
    if hv:
        ref = (F.interpolate(ref, size=(hv, wv), mode="nearest") > 0.5).float()  # resized, thresholded back to uint8
    ref = F.interpolate(ref, size=(oh, ow), mode="nearest")[0]
    got = m.index_select(1, nearest_source_index(H, hv, oh)).index_select(2, nearest_source_index(W, wv, ow)).float()
    assert torch.equal(ref, got)
