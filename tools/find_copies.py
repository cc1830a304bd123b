"""Where do layout copies (aten::clone / aten::contiguous / aten::copy_) of the eager step come from?
Aggregates device time by (op, shape, innermost repo frame)."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from dfine_seg_amd.synth import synthetic_batch
from dfine_seg_amd.train import TrainStep

t = TrainStep("m", 80, False, (640, 640), "cuda", total_steps=1000, use_graph=False)
images, targets = synthetic_batch(32, 640, 640, 80, 8, seed=42, device="cuda")
images = images.contiguous(memory_format=torch.channels_last)
for _ in range(4):
    t.step(images, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    t.step(images, targets)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if e.name in ("aten::clone", "aten::copy_", "aten::cat", "aten::add", "aten::add_", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::mul") and e.device_time_total > 0:
        frames = [f for f in (e.stack or []) if "d-fine-seg_amd" in f or "dfine_seg_amd" in f]
        where = frames[0].split("d-fine-seg_amd/")[-1] if frames else ("autograd/backward" if any("backward" in f for f in (e.stack or [])[:8]) else "?")
        key = (e.name, str(e.input_shapes)[:70], where[:70])
        agg[key][0] += e.device_time_total
        agg[key][1] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
print("total ms", sum(v[0] for v in agg.values()) / 1e3)
for (name, shapes, where), (us, n) in rows[:70]:
    print(f"{us/1e3:7.3f} ms n={n:3d} {name:14s} {shapes:70s} {where}")
