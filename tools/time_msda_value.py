import sys, time; sys.path.insert(0, "/root/repo")
import torch, ctypes
import dfine_seg_amd
from dfine_seg_amd import _lib, ops
lib = _lib.load()
B, Lq, nH, c = 32, 492, 8, 32
shapes, npts = [(80, 80), (40, 40), (20, 20)], [3, 6, 3]
S, P = 8400, 12
g = torch.Generator(device="cuda").manual_seed(0)
def run(locs, tag):
    n = len(locs)
    attn = [torch.rand(B, Lq, nH, P, device="cuda") for _ in range(n)]
    go = [torch.randn(B, Lq, nH * c, device="cuda").bfloat16() for _ in range(n)]
    out = torch.empty(B, S, nH, c, device="cuda")
    hw = _lib.i32_array(v for pair in shapes for v in pair); np_ = _lib.i32_array(npts)
    arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    def call():
        rc = lib.dfine_msda_bwd_value(1, arr(locs), arr(attn), arr(go), n, _lib.ptr(out), B, S, Lq, nH, c, 3, hw, np_, _lib.stream_ptr())
        assert rc == 0
    for _ in range(3): call()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): call()
    e.record(); torch.cuda.synchronize()
    print(tag, n, "calls:", round(s.elapsed_time(e) / 5 * 1e3, 1), "us")
inside = [torch.rand(B, Lq, nH, P, 2, device="cuda", generator=g) for _ in range(4)]
outside = [torch.full((B, Lq, nH, P, 2), 5.0, device="cuda") for _ in range(4)]
run(outside, "no hits"); run(inside, "uniform random"); run(inside[:1], "uniform random")
clustered = [(torch.rand(B, Lq, nH, 1, 2, device="cuda", generator=g) + 0.05 * torch.randn(B, Lq, nH, P, 2, device="cuda", generator=g)).clamp(0, 1).contiguous() for _ in range(4)]
run(clustered, "clustered")
