"""ATen ops of one eager step by Python call site (torch.profiler with_stack): which lines of this repo launch the glue kernels."""
import os, sys, collections
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
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.self_device_time_total <= 0:
        continue
    site = "backward (autograd engine)"
    for fr in ev.stack:
        if "d-fine-seg_amd" in fr or "dfine_seg_amd" in fr:
            site = fr.split("d-fine-seg_amd/")[-1].split("dfine_seg_amd/")[-1]
            break
    k = (ev.name, site, str(ev.input_shapes)[:90])
    agg[k][0] += ev.self_device_time_total
    agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
print(f"aten self device time {tot/1e3:.2f} ms")
for (name, site, shapes), (us, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"{us/1e3:7.3f} ms n={n:3d} {name:28s} {site:60s} {shapes}")
