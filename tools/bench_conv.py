"""Per-shape: dense conv forward / input gradient by MIOpen (F.conv2d, aten.convolution_backward) vs the
implicit-GEMM kernel (dfine_igemm_bf16), and Linear forward by hipBLASLt vs the kernel, on the shapes of
D-FINE-m 640x640 at batch 32 (tools/ are diagnostics, not part of the product path).
Usage: python tools/bench_conv.py [--no-lib] [--batch 32]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from dfine_seg_amd import _lib, gemm

ap = argparse.ArgumentParser()
ap.add_argument("--no-lib", action="store_true", help="skip the MIOpen / hipBLASLt columns")
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--only", default="")
args = ap.parse_args()
torch.backends.cudnn.benchmark = True
_lib.load()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


# (Cin, Cout, k, stride, pad, H, count per step) of D-FINE-m @ 640
CONVS = [(128, 128, 3, 1, 1, 80, 6), (128, 128, 3, 1, 1, 40, 12), (512, 512, 1, 1, 0, 80, 1), (1280, 384, 1, 1, 0, 40, 2),
         (384, 768, 1, 1, 0, 40, 3), (768, 256, 1, 1, 0, 80, 1), (32, 32, 3, 1, 1, 160, 4), (768, 256, 1, 1, 0, 40, 3),
         (512, 512, 1, 1, 0, 40, 2), (64, 64, 3, 1, 1, 80, 3), (384, 256, 1, 1, 0, 80, 1), (128, 128, 1, 1, 0, 80, 6),
         (128, 128, 1, 1, 0, 40, 21), (896, 384, 1, 1, 0, 40, 1), (1792, 768, 1, 1, 0, 20, 1), (192, 384, 1, 1, 0, 80, 1),
         (768, 1536, 1, 1, 0, 20, 1), (352, 192, 1, 1, 0, 80, 1), (256, 128, 1, 1, 0, 80, 2), (256, 256, 1, 1, 0, 80, 1),
         (96, 64, 3, 1, 1, 80, 1), (128, 128, 3, 1, 1, 20, 6), (768, 128, 1, 1, 0, 40, 2), (48, 24, 3, 2, 1, 320, 1),
         (256, 128, 1, 1, 0, 40, 4), (160, 48, 1, 1, 0, 160, 1), (48, 96, 1, 1, 0, 160, 1), (24, 32, 1, 1, 0, 160, 1),
         (3, 24, 3, 2, 1, 640, 1), (24, 12, 2, 1, 0, 320, 1), (12, 24, 2, 1, 0, 320, 1)]
B = args.batch
tot = {"mi_f": 0.0, "ig_f": 0.0, "mi_d": 0.0, "ig_d": 0.0}
print(f"conv B={B}: Cin->Cout k s @H x count | fwd: miopen us, igemm us (TFLOP/s, GB/s) | dgrad: miopen us, igemm us")
for ci, co, k, s, p, H, cnt in CONVS:
    if args.only and args.only not in f"{ci}->{co}k{k}":
        continue
    pbr = 1 if (k == 2) else None
    x = torch.randn(B, ci, H, H, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, k, k, device="cuda", dtype=torch.bfloat16) / (k * k * ci) ** 0.5).contiguous(memory_format=torch.channels_last)
    xl = x.permute(0, 2, 3, 1)
    wk = w.permute(0, 2, 3, 1).contiguous()
    wt = w.permute(1, 2, 3, 0).contiguous()
    OH, OW = gemm.conv_out_hw(H, H, k, k, s, p, pbr)
    y = torch.empty(B, OH, OW, co, device="cuda", dtype=torch.bfloat16)
    gy = torch.randn(B, co, OH, OW, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gyl = gy.permute(0, 2, 3, 1)
    dx = torch.empty(B, H, H, ci, device="cuda", dtype=torch.bfloat16)
    xp = F.pad(x, (0, 1, 0, 1)) if pbr else x
    t_mf = t_md = float("nan")
    if not args.no_lib:
        t_mf = timeit(lambda: F.conv2d(xp, w, None, s, p))
        if ci > 3 and not pbr:
            t_md = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (s, s), (p, p), (1, 1), False, (0, 0), 1, (True, False, False)))
    t_if = timeit(lambda: gemm.conv_forward(xl, wk, s, p, pbr, out=y))
    t_id = float("nan")
    if ci > 3 and not pbr:
        t_id = timeit(lambda: gemm.conv_input_grad(gyl, wt, (H, H), s, p, out=dx))
    fl = 2.0 * B * OH * OW * k * k * ci * co
    by = (x.numel() + y.numel()) * 2
    print(f"  {ci:5d}->{co:5d} k{k} s{s} @{H:3d} x{cnt:2d} | {t_mf:8.1f} {t_if:8.1f} ({fl / t_if / 1e6:5.0f} TF, {by / t_if / 1e3:5.0f} GB/s) | {t_md:8.1f} {t_id:8.1f} ({fl / t_id / 1e6:5.0f} TF)")
    tot["mi_f"] += cnt * (t_mf if t_mf == t_mf else 0); tot["ig_f"] += cnt * t_if
    tot["mi_d"] += cnt * (t_md if t_md == t_md else 0); tot["ig_d"] += cnt * (t_id if t_id == t_id else 0)
print("per-step totals (us):", {k: round(v) for k, v in tot.items()})

print("linear fwd: M N K | hipblaslt us | igemm us")
for M, N, K in [(15744, 1024, 256), (15744, 256, 1024), (15744, 256, 256), (15744, 512, 256), (15744, 512, 512), (15744, 192, 256),
                (15744, 96, 256), (15744, 80, 256), (15744, 132, 256), (268800, 256, 256), (268800, 80, 256), (12800, 1024, 256),
                (12800, 256, 1024), (12800, 512, 256), (9600, 256, 256)]:
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) / K ** 0.5
    b = torch.randn(N, device="cuda")
    b16 = b.to(torch.bfloat16)
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t_b = float("nan") if args.no_lib else timeit(lambda: F.linear(x, w, b16))
    t_i = timeit(lambda: gemm.linear_forward(x, w, b, False, out=y))
    print(f"  M={M:7d} N={N:5d} K={K:5d}: {t_b:8.1f} | {t_i:8.1f}  ({2.0 * M * N * K / t_i / 1e6:5.0f} TF)")
