"""Every weight-gradient launch of one eager D-FINE-m step, timed in place (HIP events around the call, so the caches
are in the state the step leaves them in), grouped by shape: where do the wgrad milliseconds go?"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfine_seg_amd import _lib
from dfine_seg_amd.synth import synthetic_batch
from dfine_seg_amd.train import TrainStep

lib = _lib.load()
calls = []
def wrap(name, describe):
    fn = getattr(lib, name)
    def w(*a):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); rc = fn(*a); e.record()
        calls.append((describe(a), s, e))
        return rc
    setattr(lib, name, w)
def conv_desc(a):
    B, H, W, CI, OH, OW, CO, KH, KW, st, pad = a[3:14]
    return ("conv", B * OH * OW, CO, KH * KW * CI, f"{CI}->{CO} k{KH} s{st} @{OH}")
def lin_desc(a):
    M, N, K = a[4:7]
    return ("linear", M, N, K, "bias" if a[3] else "")
wrap("dfine_conv_wgrad", conv_desc)
wrap("dfine_linear_wgrad", lin_desc)
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
    kind, M, N, K, note = d
    fl = 2.0 * M * N * K * n
    by = (M * (N + (K if kind == "linear" else K // (9 if "k3" in note else 1))) * 2 + N * K * 4) * n
    print(f"{us:8.1f} us x{n:3d} {us/n:7.1f} us  {kind:6s} M={M:7d} N={N:5d} K={K:5d} {note:24s} {fl/us/1e6:6.0f} TFLOP/s {by/us/1e3:6.0f} GB/s")
