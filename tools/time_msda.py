"""Micro-benchmark of the MSDA kernels at BASELINE shapes (HIP events on the launch stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfine_seg_amd import ops

def main():
    B, Lq, nH, c, shapes, npts = 32, 492, 8, 32, [(80, 80), (40, 40), (20, 20)], [3, 6, 3]
    S, P = sum(h * w for h, w in shapes), sum(npts)
    g = torch.Generator(device="cuda").manual_seed(0)
    for dtype in (torch.bfloat16, torch.float32):
        value = torch.randn(B, S, nH, c, generator=g, device="cuda").to(dtype).requires_grad_(True)
        ref = torch.rand(B, Lq, 1, 1, 2, generator=g, device="cuda") * 0.6 + 0.2
        loc = (ref + 0.1 * torch.randn(B, Lq, nH, P, 2, generator=g, device="cuda")).requires_grad_(True)
        attn = torch.softmax(torch.randn(B, Lq, nH, P, generator=g, device="cuda"), -1).requires_grad_(True)
        go = torch.randn(B, Lq, nH * c, generator=g, device="cuda").to(dtype)
        for _ in range(3):
            out = ops.deformable_attention_core(value, shapes, loc, attn, npts)
            out.backward(go)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        n = 20
        tf = tb = 0.0
        for _ in range(n):
            e[0].record()
            out = ops.deformable_attention_core(value, shapes, loc, attn, npts)
            e[1].record()
            out.backward(go)
            e[2].record()
            torch.cuda.synchronize()
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
        es = value.element_size()
        alg_f = B * (Lq * nH * P * 4 * c * es + Lq * nH * P * 12 + Lq * nH * c * es)
        alg_b = alg_f + B * Lq * nH * P * 4 * c * 4 + B * Lq * nH * P * 12
        print(f"{dtype}: fwd {tf/n*1e3:.1f} us ({alg_f/(tf/n*1e-3)/1e9:.0f} GB/s alg)  "
              f"bwd(autograd incl. zero+cast) {tb/n*1e3:.1f} us ({alg_b/(tb/n*1e-3)/1e9:.0f} GB/s alg)")

if __name__ == "__main__":
    main()
