"""d(loss)/d(value) of the deformable attention at D-FINE-m batch 32, 4 calls: the LDS-slab kernel (dfine_msda_bwd_value), the
sorted gather (dfine_msda_bwd_value_sorted; DFINE_MSDA_SORT_DEBUG=1: count + scan only, 2: + fill, no gather) and the float-atomic
scatter (dfine_msda_bwd, all of it), on sampling locations that miss, are uniform, or cluster (queries around 8 boxes per image)."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    cap = lib.dfine_msda_sorted_records(n, B, S, Lq, nH, P)
    entries = torch.empty(cap, 2, dtype=torch.int32, device="cuda")
    value = torch.randn(B, S, nH, c, device="cuda").bfloat16()
    gl, ga = torch.empty_like(locs[0]), torch.empty_like(attn[0])
    def slab():
        assert lib.dfine_msda_bwd_value(1, arr(locs), arr(attn), arr(go), n, _lib.ptr(out), B, S, Lq, nH, c, 3, hw, np_, _lib.stream_ptr()) == 0
    def sorted_():
        assert lib.dfine_msda_bwd_value_sorted(1, arr(locs), arr(attn), arr(go), n, _lib.ptr(out), 0, _lib.ptr(entries), cap, B, S, Lq, nH, c, 3, hw, np_,
                                               _lib.stream_ptr()) == 0
    def atomic():
        for i in range(n):
            assert lib.dfine_msda_bwd(_lib.ptr(value), 1, _lib.ptr(locs[i]), _lib.ptr(attn[i]), _lib.ptr(go[i]), _lib.ptr(out), 0, _lib.ptr(gl), _lib.ptr(ga),
                                      B, S, Lq, nH, c, 3, hw, np_, _lib.stream_ptr()) == 0
    def points():
        for i in range(n):
            assert lib.dfine_msda_bwd_points(_lib.ptr(value), 1, _lib.ptr(locs[i]), _lib.ptr(attn[i]), _lib.ptr(go[i]), _lib.ptr(gl), _lib.ptr(ga),
                                             B, S, Lq, nH, c, 3, hw, np_, _lib.stream_ptr()) == 0
    line = f"{tag:16s} {n} calls:"
    variants = [("slab", slab, {}), ("sorted", sorted_, {}), ("sorted:count", sorted_, {"DFINE_MSDA_SORT_DEBUG": "1"}),
                ("sorted:count+fill", sorted_, {"DFINE_MSDA_SORT_DEBUG": "2"}), ("atomic(all)", atomic, {}), ("points", points, {})]
    for name, call, env in variants:
        os.environ.pop("DFINE_MSDA_SORT_DEBUG", None)
        os.environ.update(env)
        for _ in range(2): call()
        torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): call()
        e.record(); torch.cuda.synchronize()
        line += f"  {name} {s.elapsed_time(e) / 5 * 1e3:7.1f}"
    os.environ.pop("DFINE_MSDA_SORT_DEBUG", None)
    print(line, "us", flush=True)
inside = [torch.rand(B, Lq, nH, P, 2, device="cuda", generator=g) for _ in range(4)]
outside = [torch.full((B, Lq, nH, P, 2), 5.0, device="cuda") for _ in range(4)]
run(outside, "no hits"); run(inside, "uniform random"); run(inside[:1], "uniform random")
clustered = [(torch.rand(B, Lq, nH, 1, 2, device="cuda", generator=g) + 0.05 * torch.randn(B, Lq, nH, P, 2, device="cuda", generator=g)).clamp(0, 1).contiguous() for _ in range(4)]
run(clustered, "clustered")
# what training does: every query sits on one of 8 boxes per image (matched + denoising queries), points spread inside the box
boxes = torch.rand(B, 8, 2, device="cuda", generator=g) * 0.7 + 0.15
pick = torch.randint(0, 8, (B, Lq), device="cuda", generator=g)
centre = boxes.gather(1, pick.unsqueeze(-1).expand(-1, -1, 2)).view(B, Lq, 1, 1, 2)
on_boxes = [(centre + 0.06 * torch.randn(B, Lq, nH, P, 2, device="cuda", generator=g)).clamp(0, 1).contiguous() for _ in range(4)]
run(on_boxes, "on 8 boxes")
