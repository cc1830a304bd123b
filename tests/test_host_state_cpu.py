"""Host-side state that the captured step depends on, checked without a GPU: the zero-initialised scratch
arena (what a frozen memset must cover) and the bf16 weight-shadow bookkeeping (staleness by tensor version)."""
import torch

import dfine_seg_amd  # noqa: F401
from dfine_seg_amd import ops
from dfine_seg_amd.optim import FlatState


def test_zero_arena_rezeroes_everything_ever_handed_out():
    arena = ops.ZeroArena(64, "cpu")
    a = arena.take(10)  # rounded up to 12 (16-byte granules)
    b = arena.take(5)
    assert a.numel() == 12 and b.numel() == 8 and arena.used == 20
    a.fill_(1.0), b.fill_(2.0)
    arena.reset()
    assert arena.used == 0 and float(arena.buf.abs().sum()) == 0.0
    # the memset covers the WHOLE arena, whatever extent this or any earlier step used: a graph captured with a short
    # step and one captured later with a longer step (another GT count, with masks) both start from zeros on replay
    arena.take(4).fill_(3.0)
    arena.buf[40:44].fill_(5.0)  # stale data beyond every extent handed out so far
    arena.reset()
    assert float(arena.buf.abs().sum()) == 0.0
    assert arena.take(100) is None  # exhausted: callers fall back to their own torch.zeros


def test_bf16_shadow_staleness_is_detected_by_version():
    model = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 4))
    flat = FlatState(model)
    assert flat.bf16_shadow_is_current()  # no shadow: nothing can be stale
    flat.enable_bf16_shadow()
    w = model[0].weight
    assert ops.bf16_of(w).data_ptr() == w._bf16_shadow.data_ptr()
    assert torch.equal(flat.shadow, flat.params.to(torch.bfloat16))
    model.load_state_dict({k: v + 1 for k, v in model.state_dict().items()})  # writes behind the optimizer's back
    assert not flat.bf16_shadow_is_current()
    assert torch.equal(ops.bf16_of(w), w.detach().to(torch.bfloat16))  # a cast of the new value, not the stale copy
    flat.refresh_bf16_shadow()
    assert flat.bf16_shadow_is_current() and torch.equal(w._bf16_shadow, w.detach().to(torch.bfloat16))
    # parameters stay views of the flat buffers with their original shapes
    assert w.data_ptr() >= flat.params.data_ptr() and w.shape == (8, 8)
