"""igemm forward on representative shapes of D-FINE-m with COLD operands: every launch of a timed graph reads / writes its
own buffers out of a rotating set larger than the 256 MB Infinity Cache, which is how the kernel runs inside the training
step (the plain per-shape loop of tools/bench_conv.py re-reads MALL-resident tensors and flatters HBM-bound shapes).
Usage: python tools/bench_igemm_cold.py [--stats]   (DFINE_IGEMM_BK=32|64 selects the K step)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfine_seg_amd import _lib, gemm, ops

ap = argparse.ArgumentParser()
ap.add_argument("--stats", action="store_true")
args = ap.parse_args()
_lib.load()
B = 32
SHAPES = [(128, 128, 1, 80), (128, 128, 1, 40), (128, 128, 3, 80), (128, 128, 3, 40), (512, 512, 1, 80), (384, 768, 1, 40), (768, 256, 1, 80),
          (1280, 384, 1, 40), (256, 128, 1, 80), (32, 32, 3, 160), (64, 64, 3, 80), (160, 48, 1, 160), (1792, 768, 1, 20), (128, 128, 3, 20)]
print(f"BK={os.environ.get('DFINE_IGEMM_BK', 'auto')} stats={args.stats}: Cin->Cout k @H | us | TF/s | GB/s (in+out)")
for ci, co, k, H in SHAPES:
    nbytes = B * H * H * (ci + co) * 2
    n_sets = max(2, int(600e6 // nbytes) + 1)
    xs = [torch.randn(B, H, H, ci, device="cuda", dtype=torch.bfloat16) for _ in range(n_sets)]
    ys = [torch.empty(B, H, H, co, device="cuda", dtype=torch.bfloat16) for _ in range(n_sets)]
    w = torch.randn(co, k, k, ci, device="cuda", dtype=torch.bfloat16) / (k * k * ci) ** 0.5
    st = [torch.zeros(32 * co, device="cuda") for _ in range(n_sets)] if args.stats else [None] * n_sets
    reps = max(n_sets, 16)
    g = torch.cuda.CUDAGraph()
    for i in range(2):
        gemm.conv_forward(xs[i], w, 1, k // 2, out=ys[i], stats=st[i])
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for i in range(reps):
            gemm.conv_forward(xs[i % n_sets], w, 1, k // 2, out=ys[i % n_sets], stats=st[i % n_sets])
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / reps * 1e3
    fl = 2.0 * B * H * H * k * k * ci * co
    print(f"  {ci:5d}->{co:5d} k{k} @{H:3d}: {us:8.1f} us  {fl / us / 1e6:6.0f} TF  {nbytes / us / 1e3:6.0f} GB/s")
