"""Fused optimizer tail (clip + AdamW + EMA + zero_grad over flat buffers) against the stock
PyTorch sequence the reference runs (src/dl/train.py:488-511, 61-72), and the OneCycle schedule
function against torch.optim.lr_scheduler.OneCycleLR."""
import copy
import math

import pytest
import torch
import torch.nn as nn

import dfine_seg_amd  # noqa: F401
from dfine_seg_amd.optim import FlatState, FusedAdamWEMA, group_parameters, one_cycle_lr


def test_one_cycle_matches_torch_scheduler():
    p = nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1e-3)
    total = 200
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=3e-4, total_steps=total, pct_start=0.1, cycle_momentum=False)
    for step in range(total):
        assert math.isclose(opt.param_groups[0]["lr"], one_cycle_lr(step, total, 3e-4, 0.1), rel_tol=1e-9, abs_tol=1e-15), step
        opt.step()
        if step + 1 < total:
            sched.step()


class Tiny(nn.Module):  # names chosen so all four reference groups are populated
    def __init__(self):
        super().__init__()
        self.backbone = nn.Sequential(nn.Conv2d(3, 8, 3), nn.BatchNorm2d(8))
        self.backbone[1].register_parameter  # noqa: B018
        self.encoder = nn.ModuleDict({"proj": nn.Linear(8, 16), "norm": nn.LayerNorm(16)})
        self.decoder = nn.Linear(16, 5)
        self.frozen = nn.Parameter(torch.ones(3), requires_grad=False)


def test_flat_state_keeps_values_shapes_and_layout():
    torch.manual_seed(0)
    model = Tiny().to(memory_format=torch.channels_last)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    strides = {n: p.stride() for n, p in model.named_parameters()}
    flat = FlatState(model)
    after = model.state_dict()
    assert list(after.keys()) == list(before.keys())
    for k in before:
        assert torch.equal(before[k], after[k])
    for n, p in model.named_parameters():
        assert p.stride() == strides[n]
        if p.requires_grad:
            assert p.grad is not None and p.grad.stride() == p.stride() and float(p.grad.abs().sum()) == 0
    assert flat.numel >= sum(p.numel() for p in model.parameters() if p.requires_grad)
    assert flat.group_begin[0] == 0 and all(b % 4 == 0 for b in flat.group_begin)
    # writing through the flat buffer is writing the parameters
    flat.params.add_(1.0)
    assert torch.allclose(model.decoder.weight, before["decoder.weight"] + 1.0)


@pytest.mark.gpu
def test_fused_adamw_ema_matches_torch(hip_lib):
    torch.manual_seed(0)
    model = Tiny().cuda()
    ref_model = copy.deepcopy(model)
    ema_model = copy.deepcopy(model)
    ref_ema = copy.deepcopy(model)
    flat = FlatState(model)
    ema_flat = flat.mirror(ema_model)
    lrs = {"backbone": 2e-3, "backbone_norm": 2e-3, "encdec_norm_bias": 1e-2, "rest": 1e-2}
    wds = {"backbone": 0.05, "backbone_norm": 0.0, "encdec_norm_bias": 0.0, "rest": 0.05}
    fused = FusedAdamWEMA(flat, lrs, wds, betas=(0.9, 0.999), clip_max_norm=0.1, ema_flat=ema_flat)
    groups = group_parameters(ref_model)
    ref_opt = torch.optim.AdamW([
        {"params": [p for _, p in groups["backbone"]], "lr": 2e-3, "weight_decay": 0.05},
        {"params": [p for _, p in groups["backbone_norm"]], "lr": 2e-3, "weight_decay": 0.0},
        {"params": [p for _, p in groups["encdec_norm_bias"] if p.requires_grad], "lr": 1e-2, "weight_decay": 0.0},
        {"params": [p for _, p in groups["rest"] if p.requires_grad], "lr": 1e-2, "weight_decay": 0.05},
    ], betas=(0.9, 0.999))
    ref_params = dict(ref_model.named_parameters())
    g = torch.Generator(device="cuda").manual_seed(1)
    for step in range(1, 5):
        decay = 0.9998 * (1 - math.exp(-step / 2000))
        for n, p in model.named_parameters():
            if p.requires_grad:
                grad = torch.randn(p.shape, device="cuda", generator=g) * (3.0 if step % 2 else 0.01)  # clipped and unclipped steps
                p.grad.copy_(grad)
                ref_params[n].grad = grad.clone()
        fused.set_schedule(lrs, decay)
        fused.step()
        torch.nn.utils.clip_grad_norm_([p for p in ref_model.parameters() if p.requires_grad], 0.1)
        ref_opt.step()
        with torch.no_grad():
            for (n, e), (_, s) in zip(ref_ema.named_parameters(), ref_model.named_parameters()):
                if s.requires_grad:
                    e.mul_(decay).add_((1 - decay) * s)
        for n, p in model.named_parameters():
            torch.testing.assert_close(p, ref_params[n], rtol=2e-5, atol=1e-6, msg=lambda m: f"{n} step {step}: {m}")
            if p.requires_grad:
                assert float(p.grad.abs().max()) == 0.0  # zero_grad is part of the fused step
        for (n, e), (_, r) in zip(ema_model.named_parameters(), ref_ema.named_parameters()):
            torch.testing.assert_close(e, r, rtol=2e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("amp", [False, True])
def test_graph_replay_matches_eager(hip_lib, amp):
    """The captured step must compute what the eager step computes (same weights, same batches); with ``amp`` the
    benchmarked path (bf16 shadow weights, implicit-GEMM / wgrad kernels, direct accumulation) is what is replayed.
    Tolerance: fp32 2e-3 (atomic summation order), bf16 2e-2."""
    from dfine_seg_amd.synth import synthetic_batch
    from dfine_seg_amd.train import TrainStep

    losses = {}
    for use_graph in (False, True):
        torch.manual_seed(0)
        t = TrainStep("n", 80, False, (320, 320), "cuda", total_steps=100, use_graph=use_graph, graph_warmup=2, amp=amp)
        out = []
        for i in range(6):
            torch.manual_seed(100 + i)  # same denoising noise stream for both runs
            images, targets = synthetic_batch(2, 320, 320, 80, 4, seed=i, device="cuda")
            out.append(float(t.step(images, targets)))
        losses[use_graph] = out
        assert (len(t._graphs) == 1) == use_graph
    tol = 2e-2 if amp else 2e-3
    for a, b in zip(losses[False], losses[True]):
        assert abs(a - b) <= tol * abs(a), (losses[False], losses[True])


@pytest.mark.gpu
def test_graph_replay_with_varying_ground_truth_counts(hip_lib):
    """Real batches have a different maximum GT count almost every step: each new (padded width, max count) key is
    captured on first sight, with NO eager pass for that key (the denoising attention mask and every other cached
    constant must be built on the device), all graphs share one memory pool, and the losses must equal the eager
    run's. bf16 path, 10 steps, GT counts 1..20 per image."""
    from dfine_seg_amd.synth import synthetic_batch
    from dfine_seg_amd.train import TrainStep

    losses, n_graphs = {}, 0
    for use_graph in (False, True):
        torch.manual_seed(0)
        t = TrainStep("n", 80, False, (320, 320), "cuda", total_steps=100, use_graph=use_graph, graph_warmup=2, amp=True)
        out = []
        for i in range(10):
            torch.manual_seed(100 + i)
            images, targets = synthetic_batch(2, 320, 320, 80, 4, seed=i, device="cuda", random_counts=True)
            out.append(float(t.step(images, targets)))
        losses[use_graph] = out
        n_graphs = max(n_graphs, len(t._graphs))
    assert n_graphs >= 4, n_graphs  # several distinct keys were captured after the warm-up
    for a, b in zip(losses[False], losses[True]):
        assert abs(a - b) <= 2e-2 * abs(a), (losses[False], losses[True])


@pytest.mark.gpu
def test_steps_without_host_sync_match_synced_steps(hip_lib):
    """Graph replay lets the host run many steps ahead of the GPU: the per-step scalars (learning rates, Adam bias
    corrections, EMA decay) must still reach the step they were computed for (ring of pinned staging buffers).
    12 replayed steps with no synchronisation vs the same steps with a sync after each: same weights."""
    from dfine_seg_amd.synth import synthetic_batch
    from dfine_seg_amd.train import TrainStep

    finals = []
    images, targets = synthetic_batch(2, 320, 320, 80, 4, seed=3, device="cuda")
    for sync_every_step in (True, False):
        torch.manual_seed(0)
        t = TrainStep("n", 80, False, (320, 320), "cuda", total_steps=20, pct_start=0.3, use_graph=True, graph_warmup=2, amp=False)
        for i in range(14):
            torch.manual_seed(100 + i)
            loss = t.step(images, targets)
            if sync_every_step:
                float(loss)
        torch.cuda.synchronize()
        finals.append((t.flat.params.clone(), t.optimizer.exp_avg_sq.clone()))
    # Adam's second moment is the most sensitive to a wrong bias correction / learning-rate slot
    torch.testing.assert_close(finals[0][1], finals[1][1], rtol=1e-3, atol=1e-12)
    torch.testing.assert_close(finals[0][0], finals[1][0], rtol=1e-3, atol=2e-5)


@pytest.mark.gpu
def test_bf16_shadow_follows_the_master_weights(hip_lib):
    """The optimizer kernel keeps the bf16 copy the compute path reads equal to bf16(fp32 master); a
    weight written behind its back is detected (tensor version) and served by a cast until refreshed."""
    from dfine_seg_amd import ops
    from dfine_seg_amd.optim import FlatState, FusedAdamWEMA

    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(16, 24), torch.nn.Linear(24, 8)).cuda()
    flat = FlatState(model)
    flat.enable_bf16_shadow()
    groups = {g: 1e-2 for g in flat.group_names}
    opt = FusedAdamWEMA(flat, groups, {g: 0.0 for g in flat.group_names})
    w = model[0].weight
    assert ops.bf16_of(w).data_ptr() == w._bf16_shadow.data_ptr()
    for _ in range(3):
        flat.grads.normal_()
        opt.step()
        torch.cuda.synchronize()
        assert torch.equal(flat.shadow, flat.params.to(torch.bfloat16))
        assert torch.equal(ops.bf16_of(w), w.detach().to(torch.bfloat16))
        assert ops.bf16_of(w).data_ptr() == w._bf16_shadow.data_ptr()  # no cast launched
    with torch.no_grad():
        w.add_(1.0)  # e.g. load_state_dict
    assert not flat.bf16_shadow_is_current()
    assert torch.equal(ops.bf16_of(w), w.detach().to(torch.bfloat16))  # served by a cast, not the stale copy
    flat.refresh_bf16_shadow()
    assert flat.bf16_shadow_is_current() and ops.bf16_of(w).data_ptr() == w._bf16_shadow.data_ptr()
    assert torch.equal(flat.shadow, flat.params.to(torch.bfloat16))


@pytest.mark.gpu
@pytest.mark.parametrize("amp", [False, True])
def test_graph_replay_matches_eager_segment(hip_lib, amp):
    """Same as above with the mask branch in the captured step (graph_masks=True): pixel decoder, mask
    embeddings, matched-rows mask logits and the dense focal/dice loss must replay to the eager losses."""
    from dfine_seg_amd.synth import synthetic_batch
    from dfine_seg_amd.train import TrainStep

    losses = {}
    for use_graph in (False, True):
        torch.manual_seed(0)
        t = TrainStep("n", 80, True, (320, 320), "cuda", total_steps=100, use_graph=use_graph, graph_warmup=2, amp=amp,
                      graph_masks=True)
        out = []
        for i in range(5):
            torch.manual_seed(100 + i)
            images, targets = synthetic_batch(2, 320, 320, 80, 4, with_masks=True, seed=i, device="cuda")
            out.append(float(t.step(images, targets)))
        losses[use_graph] = out
        assert (len(t._graphs) == 1) == use_graph
    tol = 2e-2 if amp else 2e-3
    for a, b in zip(losses[False], losses[True]):
        assert abs(a - b) <= tol * abs(a), (losses[False], losses[True])
