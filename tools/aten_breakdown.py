"""Which ATen ops (with operand shapes) the eager step still spends device time in (torch.profiler)."""
import os, sys
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    t.step(images, targets)
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_input_shape=True)
rows = sorted(rows, key=lambda r: -r.self_device_time_total)
tot = sum(r.self_device_time_total for r in rows)
print(f"total self device time {tot/1e3:.2f} ms")
for r in rows[:90]:
    print(f"{r.self_device_time_total/1e3:8.3f} ms  n={r.count:4d}  {r.key:45s} {str(r.input_shapes)[:150]}")
