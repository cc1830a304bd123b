"""1x1 conv: MIOpen (F.conv2d, channels_last bf16) vs hipBLASLt (F.linear on the NHWC view), fwd+bwd."""
import torch, torch.nn.functional as F
shapes = [  # (B, C_in, C_out, H)
    (32, 160, 48, 160), (32, 48, 96, 160), (32, 352, 192, 80), (32, 192, 384, 80), (32, 896, 384, 40), (32, 384, 768, 40),
    (32, 1280, 384, 40), (32, 384, 128, 40), (32, 128, 128, 40), (32, 1792, 768, 20), (32, 768, 1536, 20), (32, 384, 256, 80),
    (32, 512, 512, 80), (32, 768, 256, 80), (32, 256, 128, 80), (32, 128, 128, 80), (32, 512, 512, 40), (32, 768, 256, 40),
]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
tot_c = tot_l = 0
for B, ci, co, H in shapes:
    x = torch.randn(B, ci, H, H, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(co, ci, 1, 1, device="cuda", dtype=torch.bfloat16) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gy = torch.randn(B, co, H, H, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    def conv():
        y = F.conv2d(x, w); y.backward(gy); x.grad = None; w.grad = None
    w2 = w.detach().view(co, ci).clone().requires_grad_(True)
    gy2 = gy.permute(0, 2, 3, 1)
    def lin():
        y = F.linear(x.permute(0, 2, 3, 1), w2); y.backward(gy2); x.grad = None; w2.grad = None
    tc, tl = timeit(conv), timeit(lin)
    tot_c += tc; tot_l += tl
    flops = 3 * 2 * B * H * H * ci * co
    print(f"B{B} {ci:5d}->{co:5d} @{H:3d}: conv2d {tc:8.1f} us ({flops/tc/1e6:6.0f} TF)  linear {tl:8.1f} us ({flops/tl/1e6:6.0f} TF)")
print("total", tot_c, tot_l)
