"""Self-attention kernels at the decoder's shape (B 32, 8 heads, L 492, head 32), with and without the denoising mask."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfine_seg_amd import ops
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
B, L, D, H = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 492, 256, 8
qk = torch.randn(B, L, 2 * D, device="cuda").bfloat16().requires_grad_(True)
v = torch.randn(B, L, D, device="cuda").bfloat16().requires_grad_(True)
mask = torch.zeros(L, L, dtype=torch.bool, device="cuda"); mask[:192, 192:] = True; mask[192:, :192] = True
go = torch.randn(B, L, D, device="cuda").bfloat16()
for name, m in (("no mask", None), ("mask", mask)):
    f = timeit(lambda: ops.self_attention(qk, v, m, H))
    def fb():
        o = ops.self_attention(qk, v, m, H); o.backward(go); qk.grad = None; v.grad = None
    fbt = timeit(fb)
    print(f"L={L} {name:8s}: fwd {f:7.1f} us   fwd+bwd {fbt:7.1f} us")
