"""Is the split-M wgrad body fast when every workgroup has a long run of rows and the grid fills the chip? One big problem
(M rows) per shape, long chunks forced with DFINE_WGRAD_ATOMIC_W -- the regime a grouped (deferred) launch would run in."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfine_seg_amd import _lib
lib = _lib.load()
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for kind, M, N, K in [("lin", 1638400, 128, 128), ("lin", 1638400, 256, 256), ("lin", 819200, 512, 512), ("lin", 1638400, 64, 64), ("lin", 1638400, 32, 288),
                      ("conv", 32 * 160 * 160, 128, 128), ("conv", 32 * 160 * 160, 64, 64), ("conv", 32 * 160 * 160, 256, 256), ("conv", 32*160*160, 32, 32)]:
    if kind == "lin":
        dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16); x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
        dw = torch.zeros(N, K, device="cuda")
        t = timeit(lambda: lib.dfine_linear_wgrad(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw), None, M, N, K, K, N, K, _lib.stream_ptr()))
        fl, by = 2.0 * M * N * K, M * (N + K) * 2
        print(f"linear M={M} N={N} K={K}: {t:8.1f} us {fl/t/1e6:6.0f} TFLOP/s {by/t/1e3:6.0f} GB/s")
    else:
        B, H, co, ci = 32, 160, N, K
        x = torch.randn(B, H, H, ci, device="cuda", dtype=torch.bfloat16); gy = torch.randn(B, H, H, co, device="cuda", dtype=torch.bfloat16)
        dw = torch.zeros(co, 3, 3, ci, device="cuda")
        t = timeit(lambda: lib.dfine_conv_wgrad(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(dw), B, H, H, ci, H, H, co, 3, 3, 1, 1, co, ci, _lib.stream_ptr()))
        fl, by = 2.0 * M * 9 * ci * co, M * (co + ci) * 2
        print(f"conv3x3 @160 {ci}->{co}: {t:8.1f} us {fl/t/1e6:6.0f} TFLOP/s {by/t/1e3:6.0f} GB/s")
