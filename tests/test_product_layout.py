"""Host-side logic of the product package that needs no GPU: checkpoint layout (the
``state_dict`` contract) against the layout recorded from the reference,
optimizer grouping, target packing."""
import pytest
import torch

from conftest import load_golden
from oracle.gen_golden import MODEL_CASES

import dfine_seg_amd  # noqa: F401
from dfine_seg_amd.dfine import build_loss, build_model, build_optimizer
from dfine_seg_amd.targets import TargetBatch


@pytest.mark.parametrize("fname", ["model_n320", "model_s640", "model_m640", "model_m640_seg", "model_x640"])
def test_state_dict_layout_matches_reference(fname):
    size, img, _, seg = MODEL_CASES[fname]
    g = load_golden(fname)
    model = build_model(size, 80, seg, "cpu", img_size=[img, img])
    sd = model.state_dict()
    assert list(sd.keys()) == [str(n) for n in g["sd.names"]]  # same names, same order
    for name, shape, dt in zip(g["sd.names"], g["sd.shapes"], g["sd.dtypes"]):
        t = sd[str(name)]
        assert list(t.shape) == [int(s) for s in shape if s >= 0], name
        assert str(t.dtype).replace("torch.", "") == str(dt), name
    trainable = [n for n, p in model.named_parameters() if p.requires_grad]
    assert trainable == [str(n) for n in g["sd.trainable"]]


def test_optimizer_groups_follow_reference_rules():
    model = build_model("m", 80, False, "cpu")
    opt = build_optimizer(model, lr=1.5e-4, backbone_lr=2e-5, betas=(0.9, 0.999), weight_decay=1.25e-4, base_lr=1.5e-4)
    sizes = [len(g["params"]) for g in opt.param_groups]
    assert sizes == [142, 120, 242, 142]  # SURVEY.md 8(a11), measured on the reference
    assert [g["weight_decay"] for g in opt.param_groups] == [1.25e-4, 0.0, 0.0, 1.25e-4]
    assert [g["lr"] for g in opt.param_groups] == [2e-5, 2e-5, 1.5e-4, 1.5e-4]


def test_build_model_missing_checkpoint_raises():
    with pytest.raises(FileNotFoundError):
        build_model("n", 80, False, "cpu", pretrained_model_path="/nonexistent/weights.pth")


def test_build_loss_terms():
    assert build_loss("m", 80, 0.0, False).losses == ("vfl", "boxes", "local")
    assert build_loss("m", 80, 0.0, True).losses == ("vfl", "boxes", "local", "masks")


def test_target_batch_packing_ragged_and_empty():
    t = [
        {"labels": torch.tensor([3, 7]), "boxes": torch.tensor([[0.5, 0.5, 0.2, 0.2], [0.3, 0.3, 0.1, 0.1]])},
        {"labels": torch.zeros(0, dtype=torch.int64), "boxes": torch.zeros(0, 4)},
        {"labels": torch.tensor([1]), "boxes": torch.tensor([[0.6, 0.4, 0.3, 0.2]])},
    ]
    tb = TargetBatch.from_list(t, 80, "cpu")
    assert tb.num_gts == [2, 0, 1] and tb.max_gt == 2 and tb.total == 3
    assert tb.labels.tolist() == [[3, 7], [80, 80], [1, 80]]
    assert tb.valid.tolist() == [[True, True], [False, False], [True, False]]
    assert tb.counts.tolist() == [2, 0, 1]
    empty = TargetBatch.from_list([t[1]], 80, "cpu")
    assert empty.total == 0 and empty.labels.shape == (1, 1)
