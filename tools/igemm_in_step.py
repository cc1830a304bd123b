"""Every implicit-GEMM launch (conv / Linear forward and input gradient) of one eager D-FINE-m step, timed in place
(HIP events around the call), grouped by shape."""
import os, sys, collections, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfine_seg_amd import _lib, gemm
from dfine_seg_amd.synth import synthetic_batch
from dfine_seg_amd.train import TrainStep

calls = []
orig = gemm._launch
def launch(A, Bw, C, bias, stats, g, name, detail):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); orig(A, Bw, C, bias, stats, g, name, detail); e.record()
    calls.append(((name, g.M, g.N, g.K, g.gather, g.ntaps, g.CI, stats is not None), s, e))
gemm._launch = launch
name = sys.argv[1] if len(sys.argv) > 1 else "m"
t = TrainStep(name, 80, False, (640, 640), "cuda", total_steps=1000, use_graph=False)
images, targets = synthetic_batch(32, 640, 640, 80, 8, seed=42, device="cuda")
images = images.contiguous(memory_format=torch.channels_last)
for _ in range(3): t.step(images, targets)
calls.clear()
t.step(images, targets); torch.cuda.synchronize()
agg = collections.OrderedDict()
for d, s, e in calls:
    r = agg.setdefault(d, [0, 0.0]); r[0] += 1; r[1] += s.elapsed_time(e) * 1e3
tot = sum(v[1] for v in agg.values())
print(f"{len(calls)} launches, {tot/1e3:.2f} ms (event-bracketed)")
for d, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    nm, M, N, K, gather, ntaps, CI, st = d
    fl = 2.0 * M * N * K * n
    by = (M * (CI if gather else K) + M * N + N * K) * 2 * n
    print(f"{us:8.1f} us x{n:3d} {us/n:7.1f} us  {nm:10s} M={M:7d} N={N:5d} K={K:5d} {'gather' + str(ntaps) if gather else 'plain':8s} {'stats' if st else '     '} {fl/us/1e6:6.0f} TFLOP/s {by/us/1e3:6.0f} GB/s")
