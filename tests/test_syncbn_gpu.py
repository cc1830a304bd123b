"""SyncBatchNorm branch of the training step (reference src/dl/train.py:159-160: under DDP with a per-GPU batch below 4 every
BatchNorm2d is converted to nn.SyncBatchNorm; SURVEY.md collective C6). The fused norm kernels serve it: the fp64 sums of the
statistics pass (or of the producing conv's epilogue) and of the backward reduction are averaged over the ranks between the two
kernels of each direction (ops.batch_norm_act(sync=...), include/dfine_hip.h: training | R << 8).

Two ranks share the one GPU of the box and talk over gloo (RCCL refuses two ranks on one device); each holds half of a batch.
The oracle is PLAIN BatchNorm on the whole batch in one process -- what SyncBatchNorm is defined to equal: outputs and input
gradients of a rank's half, parameter gradients summed over the ranks, running statistics (unbiased with the GLOBAL count)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

_LAYER_SCRIPT = r"""
import copy, os, sys, torch, torch.nn as nn, torch.distributed as dist
sys.path.insert(0, os.getcwd())
rank, world = int(os.environ["RANK"]), 2
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
from dfine_seg_amd.layers import ConvNorm

def run(amp, k, act, lab):
    torch.manual_seed(3)
    dev = "cuda"
    layer = ConvNorm(32, 64, k, act=act, lab=lab).to(dev).to(memory_format=torch.channels_last).train()
    with torch.no_grad():
        layer.norm.weight.uniform_(0.5, 1.5); layer.norm.bias.normal_(0, 0.2)
    x_full = (torch.randn(4, 32, 12, 10, device=dev) * 1.5 + 0.3).contiguous(memory_format=torch.channels_last)
    r_full = torch.randn(4, 64, 12, 10, device=dev)
    half = slice(2 * rank, 2 * rank + 2)

    def go(mod, x, r):
        x = x.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            y = mod(x)
        (y.float() * r).sum().backward()
        return y.float().detach(), x.grad.float(), {n: p.grad.float().clone() for n, p in mod.named_parameters()}

    ref = copy.deepcopy(layer)
    y_ref, dx_ref, g_ref = go(ref, x_full, r_full)
    syn = nn.SyncBatchNorm.convert_sync_batchnorm(copy.deepcopy(layer))
    assert isinstance(syn.norm, nn.SyncBatchNorm)
    y, dx, g = go(syn, x_full[half], r_full[half])
    tol = 3e-2 if amp else 2e-4
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-12))
    assert rel(y, y_ref[half]) < tol, ("y", rel(y, y_ref[half]))
    assert rel(dx, dx_ref[half]) < tol, ("dx", rel(dx, dx_ref[half]))
    for n in g:
        t = g[n].clone()
        dist.all_reduce(t)  # gloo takes the device tensor
        if n.startswith("lab.") and amp:
            continue  # a sum over all elements of a bf16 gradient map: dominated by its rounding
        assert rel(t, g_ref[n]) < (5e-2 if amp else 5e-4), (n, rel(t, g_ref[n]))
    for n in ("running_mean", "running_var"):
        a, b = getattr(syn.norm, n), getattr(ref.norm, n)
        assert rel(a, b) < (1e-2 if amp else 1e-5), (n, rel(a, b))

for amp in (False, True):
    for k, act, lab in ((3, "relu", True), (1, "silu", False), (1, None, False)):
        run(amp, k, act, lab)
dist.barrier()
if rank == 0:
    print("SYNCBN_LAYER_OK")
"""

_STEP_SCRIPT = r"""
import os, sys, torch, torch.nn as nn, torch.distributed as dist
sys.path.insert(0, os.getcwd())
rank, world = int(os.environ["RANK"]), 2
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
from dfine_seg_amd import ddp
from dfine_seg_amd.synth import synthetic_batch
from dfine_seg_amd.train import TrainStep

torch.manual_seed(7 + rank)
t = TrainStep("n", 80, False, (320, 320), "cuda", total_steps=100, grad_reducer=ddp.FlatGradAllReduce, sync_bn=True)
n_sync = sum(1 for m in t.model.modules() if isinstance(m, nn.SyncBatchNorm))
assert n_sync > 50 and not any(type(m) is nn.BatchNorm2d for m in t.model.modules()), n_sync
assert not t.use_graph
images, targets = synthetic_batch(2, 320, 320, 80, 4, seed=11 + rank, device="cuda")   # DIFFERENT data per rank
images = images.contiguous(memory_format=torch.channels_last)
for _ in range(2):
    loss = t.step(images, targets)
assert torch.isfinite(loss)
# running statistics come from the GLOBAL batch: identical on both ranks although the ranks saw different images
worst = 0.0
for n, b in t.model.named_buffers():
    if n.endswith("running_mean") or n.endswith("running_var"):
        other = b.clone()
        dist.broadcast(other, src=0)
        worst = max(worst, float((other - b).abs().max() / b.abs().max().clamp_min(1e-6)))
assert worst < 1e-5, worst
dist.barrier()
if rank == 0:
    print("SYNCBN_STEP_OK", n_sync, float(loss))
"""


def _run_two_ranks(script, token, timeout):
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", script], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                      cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-3000:]
    assert token in outs[0][0], outs[0][0][-2000:]
    return outs[0][0]


def test_sync_batch_norm_layer_equals_plain_norm_on_the_whole_batch(hip_lib):
    _run_two_ranks(_LAYER_SCRIPT, "SYNCBN_LAYER_OK", 600)


def test_train_step_with_sync_batch_norm_two_ranks(hip_lib):
    _run_two_ranks(_STEP_SCRIPT, "SYNCBN_STEP_OK", 900)
