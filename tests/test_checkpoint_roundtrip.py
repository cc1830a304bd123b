"""Checkpoint compatibility with the reference, run where the reference is present (the build
container; skipped on the GPU box): a state_dict written by this package loads strict=True into the reference's
build_model and reproduces tensor for tensor, and a file written by the reference (bare state_dict, {"model": ...} and
{"ema": {"module": ...}} layouts of src/d_fine/utils.py:156-181) loads into this package's model."""
import os

import pytest
import torch

REFERENCE = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "src", "d_fine")), reason="reference tree not present")


@pytest.mark.parametrize("size,seg", [("n", False), ("m", True)])
def test_state_dict_round_trip_with_the_reference(tmp_path, size, seg):
    from oracle import weights
    from oracle.gen_golden import import_reference

    import_reference()
    from src.d_fine.dfine import build_model as ref_build_model

    from dfine_seg_amd import checkpoint
    from dfine_seg_amd.dfine import build_model, load_checkpoint_state

    img = 320 if size == "n" else 640
    ours = build_model(size, 80, seg, "cpu", img_size=[img, img])
    ours.load_state_dict(weights.fill_state_dict(ours.state_dict()), strict=True)
    # product -> file -> reference (strict)
    checkpoint.save_model(ours, tmp_path)
    assert (tmp_path / "last.pt").exists() and (tmp_path / "model.pt").exists()
    state = torch.load(tmp_path / "model.pt", map_location="cpu", weights_only=True)
    ref = ref_build_model(size, 80, seg, "cpu", img_size=[img, img], pretrained_model_path=None)
    ref.load_state_dict(state, strict=True)
    for k, v in ref.state_dict().items():
        assert v.dtype == state[k].dtype and torch.equal(v, state[k]), k
    # reference -> file (three layouts) -> product
    with torch.no_grad():
        for p in ref.parameters():
            if p.dtype.is_floating_point:
                p.add_(0.25)
    ref_sd = ref.state_dict()
    for name, payload in [("bare", ref_sd), ("model", {"model": ref_sd}), ("ema", {"ema": {"module": ref_sd}})]:
        path = tmp_path / f"ref_{name}.pt"
        torch.save(payload, path)
        fresh = build_model(size, 80, seg, "cpu", img_size=[img, img])
        load_checkpoint_state(fresh, str(path))
        got = fresh.state_dict()
        assert list(got.keys()) == list(ref_sd.keys())
        for k in ref_sd:
            assert torch.equal(got[k], ref_sd[k]), (name, k)


def test_train_step_saves_the_ema_weights(tmp_path):
    from dfine_seg_amd import checkpoint
    from dfine_seg_amd.train import TrainStep

    t = TrainStep("n", 80, False, (320, 320), "cpu", use_graph=False, amp=False)
    with torch.no_grad():
        t.optimizer.ema_flat.add_(1.0)  # make EMA != live
    best = checkpoint.save_model(t, tmp_path, metrics={"f1": 0.5}, best_metric=0.4)
    assert best == 0.5
    saved = torch.load(tmp_path / "model.pt", map_location="cpu", weights_only=True)
    live = t.model.state_dict()
    assert list(saved.keys()) == list(live.keys())
    name = "encoder.input_proj.0.conv.weight"
    torch.testing.assert_close(saved[name], live[name] + 1.0)
    assert checkpoint.save_model(t, tmp_path, metrics={"f1": 0.3}, best_metric=best) == best  # no improvement: model.pt kept
