"""Per-shape: 1x1-conv weight gradient by MIOpen (aten.convolution_backward, weight only, incl. its helper
kernels) vs the split-M MFMA kernel; and Linear wgrad by hipBLASLt (dY^T @ X) vs the kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfine_seg_amd import _lib
lib = _lib.load()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
shapes = [(32,160,48,160),(32,48,96,160),(32,352,192,80),(32,192,384,80),(32,896,384,40),(32,384,768,40),(32,1280,384,40),
          (32,384,128,40),(32,128,128,40),(32,1792,768,20),(32,768,1536,20),(32,256,256,20),(32,384,256,80),(32,768,256,40),(32,1536,256,20),
          (32,512,512,80),(32,768,256,80),(32,256,128,80),(32,128,128,80),(32,512,512,40),(32,256,128,40),(32,256,256,80),(32,256,256,40)]
print("1x1 conv wgrad: B Cin->Cout @H : miopen us | hip us")
for B, ci, co, H in shapes:
    x = torch.randn(B, ci, H, H, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 1, 1, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, co, H, H, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw = torch.zeros(co, ci, device="cuda")
    def mi():
        return torch.ops.aten.convolution_backward(gy, x, w, None, (1,1), (0,0), (1,1), False, (0,0), 1, (False, True, False))[1].float()
    M = B*H*H
    xl, gl = x.permute(0,2,3,1), gy.permute(0,2,3,1)
    def hip():
        lib.dfine_linear_wgrad(_lib.ptr(gl), _lib.ptr(xl), _lib.ptr(dw), None, M, co, ci, ci, co, ci, _lib.stream_ptr())
    tm, th = timeit(mi), timeit(hip)
    tf = 2.0 * M * ci * co / th / 1e6
    print(f"  {ci:5d}->{co:5d} @{H:3d} M={M:7d}: {tm:8.1f} | {th:8.1f}   {'HIP' if th < tm else 'miopen'}  ({tf:.0f} TFLOP/s)")
print("3x3 conv wgrad: Cin->Cout @H stride : miopen us | hip us")
for B, ci, co, H, st in [(32,32,32,160,1),(32,32,64,160,1),(32,64,64,160,1),(32,48,48,160,1),(32,96,96,80,1),(32,64,64,80,1),(32,128,128,80,1),(32,128,128,40,1),(32,128,128,20,1),
                         (32,256,256,80,1),(32,256,256,40,1),(32,256,256,20,1),(32,32,64,160,2),(32,64,64,40,1),(32,192,192,40,1)]:
    x = torch.randn(B, ci, H, H, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(co, ci, 3, 3, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    OH = (H + 2 - 3) // st + 1
    gy = torch.randn(B, co, OH, OH, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw = torch.zeros(co, 3, 3, ci, device="cuda")
    def mi():
        return torch.ops.aten.convolution_backward(gy, x, w, None, (st,st), (1,1), (1,1), False, (0,0), 1, (False, True, False))[1].float()
    xl, gl = x.permute(0,2,3,1), gy.permute(0,2,3,1)
    def hip():
        lib.dfine_conv_wgrad(_lib.ptr(gl), _lib.ptr(xl), _lib.ptr(dw), B, H, H, ci, OH, OH, co, 3, 3, st, 1, co, ci, _lib.stream_ptr())
    tm, th = timeit(mi), timeit(hip)
    tf = 2.0 * B * OH * OH * 9 * ci * co / th / 1e6
    print(f"  {ci:5d}->{co:5d} @{H:3d} s{st}: {tm:8.1f} | {th:8.1f}   {'HIP' if th < tm else 'miopen'}  ({tf:.0f} TFLOP/s)")
print("linear wgrad: M N K : hipblaslt us | hip us")
for M, N, K in [(15744,1024,256),(15744,256,1024),(15744,256,256),(15744,512,512),(15744,192,256),(15744,96,256),(15744,80,256),(15744,136,256),(268800,256,256),(268800,80,256),(12800,1024,256),(12800,256,1024),(12800,768,256)]:
    dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16); x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    dw = torch.zeros(N, K, device="cuda")
    def bl(): return (dy.t() @ x).float()
    def hip(): lib.dfine_linear_wgrad(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw), None, M, N, K, K, N, K, _lib.stream_ptr())
    tb, th = timeit(bl), timeit(hip)
    print(f"  M={M:7d} N={N:5d} K={K:5d}: {tb:8.1f} | {th:8.1f}   {'HIP' if th < tb else 'blaslt'}")
