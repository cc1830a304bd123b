"""ATen ops of one eager step by Python call site (torch.profiler with_stack): which lines of this repo launch the glue kernels."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from dfine_seg_amd.synth import synthetic_batch
from dfine_seg_amd.train import TrainStep

seg = "--segment" in sys.argv  # python tools/aten_stacks.py [--segment]
t = TrainStep("m", 80, seg, (640, 640), "cuda", total_steps=1000, use_graph=False)
images, targets = synthetic_batch(32, 640, 640, 80, 8, with_masks=seg, seed=42, device="cuda")
images = images.contiguous(memory_format=torch.channels_last)
for _ in range(4):
    t.step(images, targets)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    t.step(images, targets)
    torch.cuda.synchronize()
# backward ops carry no Python stack: label them with the autograd node that was being evaluated when they were launched
evs = list(prof.events())
nodes = sorted(((e.time_range.start, e.time_range.end, e.name.split(": ", 1)[-1], e.thread) for e in evs
                if e.name.startswith("autograd::engine::evaluate_function")), key=lambda t: t[0])
import bisect
starts = [n[0] for n in nodes]
def node_of(e):
    i = bisect.bisect_right(starts, e.time_range.start) - 1
    while i >= 0 and i > bisect.bisect_right(starts, e.time_range.start) - 50:
        s0, e0, name, th = nodes[i]
        if s0 <= e.time_range.start and e.time_range.end <= e0 and th == e.thread:
            return name
        i -= 1
    return None
agg = collections.defaultdict(lambda: [0.0, 0])
for ev in evs:
    if not ev.name.startswith("aten::") or ev.self_device_time_total <= 0:
        continue
    site = node_of(ev)
    if site is None:
        site = "forward"
        for fr in ev.stack:
            if "d-fine-seg_amd" in fr or "dfine_seg_amd" in fr:
                site = "fwd " + fr.split("d-fine-seg_amd/")[-1].split("dfine_seg_amd/")[-1][:50]
                break
    k = (ev.name, site, str(ev.input_shapes)[:70])
    agg[k][0] += ev.self_device_time_total
    agg[k][1] += 1
tot = sum(v[0] for v in agg.values())
print(f"aten self device time {tot/1e3:.2f} ms")
for (name, site, shapes), (us, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:90]:
    print(f"{us/1e3:7.3f} ms n={n:3d} {name:26s} {site:44s} {shapes}")
