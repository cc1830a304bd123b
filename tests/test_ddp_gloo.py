"""Bucketed gradient reducer on the CPU with gloo, world_size 2 (the N>1 path of bench.py; RCCL
takes the same code path with backend "nccl")."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

import dfine_seg_amd  # noqa: F401
from dfine_seg_amd.ddp import BucketedGradReducer, FlatGradAllReduce, assign_buckets
from dfine_seg_amd.optim import FlatState


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 16)
        self.b = nn.Linear(16, 16)
        self.unused = nn.Embedding(4, 16)  # never touched in forward: its gradient must count as zero
        self.c = nn.Linear(16, 2)
        self.frozen = nn.Linear(2, 2)
        for p in self.frozen.parameters():
            p.requires_grad_(False)

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, result):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)  # ranks start from different weights: the broadcast must align them
        model = Toy()
        reducer = BucketedGradReducer(model, num_buckets=3)
        gathered = [torch.zeros_like(model.a.weight) for _ in range(world)]
        dist.all_gather(gathered, model.a.weight.detach())
        assert all(torch.equal(gathered[0], g) for g in gathered)

        opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.1)
        for step in range(3):
            x = torch.randn(5, 8, generator=torch.Generator().manual_seed(step * 10 + rank))
            model(x).pow(2).sum().backward()
            local = {n: p.grad.clone() for n, p in model.named_parameters() if p.requires_grad}
            # what the mean over ranks must be, computed with plain collectives on copies
            expect = {}
            for n, g in local.items():
                t = g.clone()
                dist.all_reduce(t)
                expect[n] = t / world
            reducer.finish()
            for n, p in model.named_parameters():
                if p.requires_grad:
                    torch.testing.assert_close(p.grad, expect[n], rtol=1e-6, atol=1e-7)
            assert float(model.unused.weight.grad.abs().max()) == 0.0
            opt.step()
            reducer.zero_grad()
            assert all(float(p.grad.abs().max()) == 0.0 for p in model.parameters() if p.requires_grad)
        gathered = [torch.zeros_like(model.c.weight) for _ in range(world)]
        dist.all_gather(gathered, model.c.weight.detach())
        assert all(torch.allclose(gathered[0], g) for g in gathered)  # replicas stayed in lock-step
        result[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_bucketed_reducer_world2_gloo():
    world = 2
    with mp.Manager() as mgr:
        result = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), result), nprocs=world, join=True)
        assert dict(result) == {0: "ok", 1: "ok"}


def _flat_worker(rank, world, port, result):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)
        model = Toy()
        reducer = FlatGradAllReduce(model, num_chunks=2)
        gathered = [torch.zeros_like(model.a.weight) for _ in range(world)]
        dist.all_gather(gathered, model.a.weight.detach())
        assert all(torch.equal(gathered[0], g) for g in gathered)  # broadcast from rank 0
        assert len(reducer.chunks) == 2 and sum(c.numel() for c in reducer.chunks) == reducer.flat.numel
        for step in range(2):
            x = torch.randn(5, 8, generator=torch.Generator().manual_seed(step * 10 + rank))
            model(x).pow(2).sum().backward()
            expect = {}
            for n, p in model.named_parameters():
                if p.requires_grad:
                    t = p.grad.clone()
                    dist.all_reduce(t)
                    expect[n] = t / world
            reducer.reduce()
            for n, p in model.named_parameters():
                if p.requires_grad:
                    torch.testing.assert_close(p.grad, expect[n], rtol=1e-6, atol=1e-7)
            reducer.zero_grad()
        result[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_flat_allreduce_world2_gloo():
    world = 2
    with mp.Manager() as mgr:
        result = mgr.dict()
        mp.spawn(_flat_worker, args=(world, _free_port(), result), nprocs=world, join=True)
        assert dict(result) == {0: "ok", 1: "ok"}


def _staged_worker(rank, world, port, result):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dfine_seg_amd.ddp import StagedGradAllReduce

        torch.manual_seed(100 + rank)
        model = Toy()
        reducer = StagedGradAllReduce(model, stage_prefixes=("c.", "b.", "a.", "unused."))  # backward order of the toy
        assert [len(r) for r in reducer.stage_ranges] == [1, 1, 1, 1]
        # every trainable element lies in exactly one stage range (ranges also span the zero alignment padding between tensors)
        assert sum(t.numel() for rs in reducer.stage_ranges for t in rs) >= sum(p.numel() for p in model.parameters() if p.requires_grad)
        for step in range(2):
            x = torch.randn(5, 8, generator=torch.Generator().manual_seed(step * 10 + rank))
            model(x).pow(2).sum().backward()
            expect = {}
            for n, p in model.named_parameters():
                if p.requires_grad:
                    t = p.grad.clone()
                    dist.all_reduce(t)
                    expect[n] = t / world
            for stage in range(4):  # stage by stage, as TrainStep drives it between its backward graphs
                reducer.reduce_stage(stage)
                done = reducer.stage_prefixes[: stage + 1]
                for n, p in model.named_parameters():
                    if p.requires_grad and n.startswith(done):
                        torch.testing.assert_close(p.grad, expect[n], rtol=1e-6, atol=1e-7)
            reducer.join()
            reducer.zero_grad()
        result[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_staged_allreduce_world2_gloo():
    world = 2
    with mp.Manager() as mgr:
        result = mgr.dict()
        mp.spawn(_staged_worker, args=(world, _free_port(), result), nprocs=world, join=True)
        assert dict(result) == {0: "ok", 1: "ok"}


def _shadow_worker(rank, world, port, result):
    """Ranks start from DIFFERENT weights with the bf16 shadow already built (TrainStep's order of construction): after
    either reducer's broadcast the shadow the compute path reads must be bf16(rank 0's weights) on every rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dfine_seg_amd import ops

        for make in (lambda f: BucketedGradReducer(f, num_buckets=2), lambda f: FlatGradAllReduce(f)):
            torch.manual_seed(100 + rank)
            model = Toy()
            flat = FlatState(model)
            flat.enable_bf16_shadow()
            before = flat.shadow.clone()
            make(flat)
            ops.DIRECT_WGRAD_ACCUM = True
            assert flat.bf16_shadow_is_current()
            assert torch.equal(flat.shadow, flat.params.to(torch.bfloat16)), "shadow not refreshed after the broadcast"
            assert torch.equal(ops.bf16_of(model.a.weight), model.a.weight.detach().to(torch.bfloat16))
            gathered = [torch.zeros_like(flat.shadow) for _ in range(world)]
            dist.all_gather(gathered, flat.shadow)
            assert all(torch.equal(gathered[0], g) for g in gathered)
            if rank > 0:
                assert not torch.equal(before, flat.shadow)  # it really was this rank's own random init before
        result[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_bf16_shadow_is_refreshed_by_the_reducers_broadcast_world2_gloo():
    world = 2
    with mp.Manager() as mgr:
        result = mgr.dict()
        mp.spawn(_shadow_worker, args=(world, _free_port(), result), nprocs=world, join=True)
        assert dict(result) == {0: "ok", 1: "ok"}


def _ema_worker(rank, world, port, result):
    """TrainStep built under a process group from per-rank seeds: live weights, EMA weights and EMA buffers must all be
    rank 0's on every rank (the EMA copy is taken before the reducer's broadcast and re-seeded after it)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dfine_seg_amd import ops
        from dfine_seg_amd.train import TrainStep

        torch.manual_seed(42 + rank)
        t = TrainStep("n", 80, False, (320, 320), "cpu", grad_reducer=FlatGradAllReduce, use_graph=False, amp=True)
        ops.DIRECT_WGRAD_ACCUM = True
        if rank == 0:  # make the buffers differ too (running statistics)
            pass
        for name, tensor in [("params", t.flat.params), ("ema", t.optimizer.ema_flat)]:
            gathered = [torch.zeros_like(tensor) for _ in range(world)]
            dist.all_gather(gathered, tensor)
            assert all(torch.equal(gathered[0], g) for g in gathered), f"{name} differ across ranks"
        assert torch.equal(t.optimizer.ema_flat, t.flat.params)
        assert torch.equal(t.flat.shadow, t.flat.params.to(torch.bfloat16))
        live = dict(t.model.named_buffers())
        for n, b in t.ema.model.named_buffers():
            assert torch.equal(b, live[n]), n
        result[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_train_step_ema_and_shadow_are_aligned_across_ranks_world2_gloo():
    world = 2
    with mp.Manager() as mgr:
        result = mgr.dict()
        mp.spawn(_ema_worker, args=(world, _free_port(), result), nprocs=world, join=True)
        assert dict(result) == {0: "ok", 1: "ok"}


def _syncbn_worker(rank, world, port, result):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dfine_seg_amd import ops
        from dfine_seg_amd.layers import ConvNorm, sync_group_of

        # the conversion the reference applies under DDP with a per-GPU batch below 4 (src/dl/train.py:159-160) reaches the
        # norm inside this repo's ConvNorm, parameters kept; the fused path then shares statistics over (default group, 2 ranks)
        layer = nn.SyncBatchNorm.convert_sync_batchnorm(ConvNorm(4, 8, 3, act="relu"))
        assert isinstance(layer.norm, nn.SyncBatchNorm) and sync_group_of(layer.norm) == (None, world)
        assert sync_group_of(nn.SyncBatchNorm(8, process_group=dist.new_group([rank]))) is None  # one rank: nothing to share
        # the fp64 sums of a norm workspace travel as their fp32 view: the average must be taken on the doubles
        sums = torch.tensor([1.0 + rank, 1e-9 * (3 + rank), 12345.678 * (rank + 1)], dtype=torch.float64)
        ws = sums.clone().view(torch.float32)
        ops._average_over_ranks(ws, (None, world))
        expect = torch.tensor([1.5, 3.5e-9, 12345.678 * 1.5], dtype=torch.float64)
        torch.testing.assert_close(ws.view(torch.float64), expect, rtol=1e-15, atol=0)
        result[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_sync_batch_norm_group_and_fp64_sum_average_world2_gloo():
    world = 2
    with mp.Manager() as mgr:
        result = mgr.dict()
        mp.spawn(_syncbn_worker, args=(world, _free_port(), result), nprocs=world, join=True)
        assert dict(result) == {0: "ok", 1: "ok"}


def test_bucket_assignment_covers_every_trainable_parameter_once():
    model = Toy()
    buckets = assign_buckets(FlatState(model).entries, num_buckets=3, min_bucket_bytes=64)
    names = [e[0] for b in buckets for e in b]
    trainable = [n for n, p in model.named_parameters() if p.requires_grad]
    assert sorted(names) == sorted(trainable) and len(names) == len(set(names))
    assert names[0].startswith("c.")  # last registered first: backward produces it first
    assert 1 < len(buckets) <= 3


def _norm_worker(rank, world, port, result):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dfine_seg_amd.criterion import DFINECriterion

        state = {"norm": torch.tensor([10.0 + rank, 4.0 * (rank + 1)])}  # [matched queries of the GO union, ground truths]
        DFINECriterion.reduce_normalisers(state)
        want = torch.tensor([sum(10.0 + r for r in range(world)) / world, sum(4.0 * (r + 1) for r in range(world)) / world])
        torch.testing.assert_close(state["norm"], want)
        result[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_criterion_normalisers_are_averaged_over_ranks_gloo():
    """num_boxes / num_boxes_go = sum over ranks / world size (dfine_criterion.py:436-453), as one message."""
    world = 2
    with mp.Manager() as mgr:
        result = mgr.dict()
        mp.spawn(_norm_worker, args=(world, _free_port(), result), nprocs=world, join=True)
        assert dict(result) == {0: "ok", 1: "ok"}


def test_criterion_normaliser_allreduce_is_single_message():
    """The product criterion fuses num_boxes / num_boxes_go into ONE 2-float all-reduce
    (reference: two all-reduces + two .item(), dfine_criterion.py:436-453)."""
    import inspect

    from dfine_seg_amd import criterion

    src = inspect.getsource(criterion.DFINECriterion)
    assert src.count("dist.all_reduce(") == 1


_RCCL_SCRIPT = """
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
import dfine_seg_amd
from dfine_seg_amd.ddp import BucketedGradReducer
from dfine_seg_amd.synth import synthetic_batch
from dfine_seg_amd.train import TrainStep
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT={port!r}, RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
trainer = TrainStep("n", 80, False, (320, 320), "cuda", grad_reducer=BucketedGradReducer, total_steps=100)
images, targets = synthetic_batch(2, 320, 320, 80, 4, seed=1, device="cuda")
for _ in range(3):
    loss = trainer.step(images, targets)
assert torch.isfinite(loss)
assert all(torch.isfinite(p).all() for p in trainer.model.parameters())
assert float(trainer.flat.grads.abs().max()) == 0.0  # the fused optimizer zeroed the reduced gradients
dist.destroy_process_group()
print("RCCL_OK")
"""


@pytest.mark.gpu
def test_reducer_rccl_single_rank_train_step(hip_lib):
    """The RCCL code path (backend "nccl", side stream, events, flat buckets) with a 1-rank group on the
    one GPU of the test box: three full training steps of D-FINE-n must run and stay finite. Run the way
    bench.py runs it -- a fresh process whose first act is init_process_group (the process group must
    exist before autograd's device thread does; pytest's earlier GPU tests would have started it)."""
    import subprocess
    import sys

    from conftest import ROOT

    script = _RCCL_SCRIPT.format(root=ROOT, port=str(_free_port()))
    proc = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    err = "\n".join(l for l in proc.stderr.splitlines() if not l.startswith("frame #"))
    assert proc.returncode == 0 and "RCCL_OK" in proc.stdout, proc.stdout[-2000:] + err[-6000:]


_RCCL_GRAPH_SCRIPT = """
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
import dfine_seg_amd
from dfine_seg_amd.ddp import FlatGradAllReduce, StagedGradAllReduce
from dfine_seg_amd.synth import synthetic_batch
from dfine_seg_amd.train import TrainStep, _SplitGraph, _StagedGraph
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT={port!r}, RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
images, targets = synthetic_batch(2, 320, 320, 80, 4, seed=1, device="cuda")
losses = []
for reducer in (None, FlatGradAllReduce, StagedGradAllReduce):
    torch.manual_seed(0)
    trainer = TrainStep("n", 80, False, (320, 320), "cuda", grad_reducer=reducer, total_steps=100, graph_warmup=2, amp=False)
    out = []
    for i in range(5):
        torch.manual_seed(10 + i)
        out.append(float(trainer.step(images, targets)))
    losses.append(out)
    assert len(trainer._graphs) == 1
    kind = {{None: torch.cuda.CUDAGraph, FlatGradAllReduce: _SplitGraph, StagedGradAllReduce: _StagedGraph}}[reducer]
    assert isinstance(next(iter(trainer._graphs.values()))[0], kind)
    assert float(trainer.flat.grads.abs().max()) == 0.0
    del trainer
    import gc; gc.collect(); torch.cuda.synchronize()
for other in losses[1:]:
    for i, (a, b) in enumerate(zip(losses[0], other)):
        # fp32 steps; float-atomic order is the only run-to-run difference, but it is amplified step over step (a Hungarian
        # assignment can flip on a round-off): tight on the first steps, loose on the last
        assert abs(a - b) <= (2e-3 if i < 3 else 2e-2) * abs(a), losses
dist.destroy_process_group()
print("RCCL_GRAPH_OK")
"""


@pytest.mark.gpu
def test_flat_allreduce_between_captured_graphs_rccl_single_rank(hip_lib):
    """bench.py's N>1 schedules on the one GPU of the test box (1-rank RCCL group): the flat schedule (graph(forward+backward)
    -> flat all-reduce -> graph(optimizer)) and the default staged schedule (backward cut into decoder / encoder / backbone
    graphs, each stage's gradient ranges all-reduced on a side stream under the next stage) must follow the same loss
    trajectory as the single-GPU graph."""
    import subprocess
    import sys

    from conftest import ROOT

    script = _RCCL_GRAPH_SCRIPT.format(root=ROOT, port=str(_free_port()))
    proc = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    err = "\n".join(l for l in proc.stderr.splitlines() if not l.startswith("frame #"))
    assert proc.returncode == 0 and "RCCL_GRAPH_OK" in proc.stdout, proc.stdout[-2000:] + err[-6000:]


_RCCL_GRAPH_AMP_SCRIPT = """
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import dfine_seg_amd
from dfine_seg_amd.ddp import FlatGradAllReduce, StagedGradAllReduce
from dfine_seg_amd.synth import synthetic_batch
from dfine_seg_amd.train import TrainStep, _SplitGraph, _StagedGraph
from test_optim import _snapshot, _restore
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT={port!r}, RANK="0", WORLD_SIZE="1")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
batches = [synthetic_batch(2, 320, 320, 80, n, seed=1 + n, device="cuda") for n in (4, 6)]  # two GT-count keys = two graphs
rel = lambda a, b: float((a - b).norm() / b.norm())
for reducer in (None, FlatGradAllReduce, StagedGradAllReduce):
    name = getattr(reducer, "__name__", "unsplit")
    torch.manual_seed(0)
    trainer = TrainStep("n", 80, False, (320, 320), "cuda", grad_reducer=reducer, total_steps=100, graph_warmup=1, amp=True)
    # the flat gradient as the optimizer finds it: a copy made at the head of the optimizer tail (captured with it)
    snap_g = torch.zeros_like(trainer.flat.grads)
    launch = trainer.optimizer.launch
    def snapshotting_launch(snap_g=snap_g, launch=launch, flat=trainer.flat):
        snap_g.copy_(flat.grads)
        launch()
    trainer.optimizer.launch = snapshotting_launch
    torch.manual_seed(5)
    trainer.step(*batches[0])  # the eager warm-up step (moves the weights off their initial values)
    for key, batch in enumerate(batches):
        state = _snapshot(trainer)
        runs = []
        for mode in ("eager", "capture+replay", "replay", "replay"):  # all FROM THE SAME STATE: only atomic order differs
            _restore(trainer, state)
            trainer.use_graph = mode != "eager"
            torch.manual_seed(200 + key)
            loss = float(trainer.step(*batch))
            assert trainer.last_step_replayed == (mode != "eager"), (name, mode)
            runs.append((loss, snap_g.clone(), trainer.flat.params.clone()))
        for mode, (loss, grad, params) in zip(("capture+replay", "replay", "replay"), runs[1:]):
            cos = float(torch.dot(grad, runs[0][1]) / (grad.norm() * runs[0][1].norm()))
            print(name, "key", key, mode, "loss", loss, "vs eager", runs[0][0], "gradient cosine", round(cos, 5), "norm ratio",
                  round(float(grad.norm() / runs[0][1].norm()), 4), "params rel", rel(params, runs[0][2]), file=sys.stderr)
            assert abs(loss - runs[0][0]) <= 1e-2 * abs(runs[0][0]), (name, key, mode, loss, runs[0][0])
            assert cos > 0.995 and abs(float(grad.norm() / runs[0][1].norm()) - 1) < 0.03, (name, key, mode, cos)
            assert rel(params, runs[0][2]) < 3e-3, (name, key, mode)
    assert len(trainer._graphs) == 2
    kind = {{None: torch.cuda.CUDAGraph, FlatGradAllReduce: _SplitGraph, StagedGradAllReduce: _StagedGraph}}[reducer]
    assert all(isinstance(e[0], kind) for e in trainer._graphs.values())
    assert trainer.wgrad_queue.launches > 0 and trainer.wgrad_queue.single_fallbacks == 0, (trainer.wgrad_queue.launches, trainer.wgrad_queue.single_fallbacks)
    assert float(trainer.flat.grads.abs().max()) == 0.0
    del trainer, launch, snapshotting_launch
    import gc; gc.collect(); torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_GRAPH_AMP_OK")
"""


@pytest.mark.gpu
def test_staged_and_flat_graph_schedules_bf16_rccl_single_rank(hip_lib):
    """The N>1 schedules in the precision that is benchmarked (amp=True: implicit-GEMM kernels, ConcatBuffer gradients parked
    across the detach cuts, WgradQueue.flush() per captured stage with persistent launch tables, bf16 leaf gradients chained by
    _backward_through) on a 1-rank RCCL group. For the unsplit graph, the flat and the staged schedule, and for TWO
    ground-truth-count keys each: the step run eagerly and its capture + 3 replays all start FROM THE SAME STATE (weights,
    moments, EMA, BatchNorm statistics, counters, RNG), so they differ only by float-atomic order -- loss within 1e-2, the flat
    gradient the optimizer graph receives at cosine > 0.995 and norm within 3 %, updated weights within 3e-3. Every weight
    gradient goes through the grouped launch (no single-launch fallback)."""
    import subprocess
    import sys

    from conftest import ROOT

    script = _RCCL_GRAPH_AMP_SCRIPT.format(root=ROOT, port=str(_free_port()))
    proc = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=900)
    err = "\n".join(l for l in proc.stderr.splitlines() if not l.startswith("frame #"))
    assert proc.returncode == 0 and "RCCL_GRAPH_AMP_OK" in proc.stdout, proc.stdout[-2000:] + err[-6000:]
