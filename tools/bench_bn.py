"""The three BatchNorm streaming kernels (apply, backward reduce, backward dx) alone, on the model's map shapes with COLD
operands (rotating buffer sets larger than the Infinity Cache inside one hipGraph), next to a plain device copy of the same
bytes. Knobs are read per call by normact.hip: DFINE_BN_REDUCE_BLOCKS, DFINE_BN_APPLY_BLOCKS, DFINE_BN_BWD_UNROLL.
Usage: python tools/bench_bn.py [knob=value ...]   (several settings: run it once per setting, or use --sweep)"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dfine_seg_amd import _lib

lib = _lib.load()
SHAPES = [(3276800, 24, "relu", True), (819200, 32, "relu", True), (204800, 128, "relu", True), (204800, 128, "silu", False),
          (51200, 128, "silu", False), (51200, 768, "relu", True), (12800, 128, "silu", False), (12800, 256, "silu", False)]
ACT = {None: 0, "relu": 1, "silu": 2}


def timed_graph(fn, n_sets, reps):
    for i in range(2):
        fn(i % n_sets)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            fn(i % n_sets)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps * 1e3)
    return best


def run(label):
    print(f"--- {label}")
    for n_pix, C, act, lab in SHAPES:
        S = n_pix * C * 2
        n_sets = max(2, int(700e6 // (3 * S)) + 1)
        reps = max(n_sets, 12)
        x = [torch.randn(n_pix, C, device="cuda").to(torch.bfloat16) for _ in range(n_sets)]
        dy = [torch.randn(n_pix, C, device="cuda").to(torch.bfloat16) for _ in range(n_sets)]
        y = [torch.empty(n_pix, C, device="cuda", dtype=torch.bfloat16) for _ in range(n_sets)]
        gamma, beta = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        ls, lb = (torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")) if lab else (None, None)
        ws = [torch.zeros(34 * C, device="cuda") for _ in range(n_sets)]
        red = [torch.zeros(66 * C + 2, device="cuda") for _ in range(n_sets)]
        st = _lib.stream_ptr

        def fwd(i, phase):
            w = ws[i]
            return lib.dfine_bn_act_fwd(_lib.ptr(x[i]), None, _lib.ptr(y[i]), 1, _lib.ptr(w), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(rm), _lib.ptr(rv),
                                        _lib.ptr(w[32 * C:]), _lib.ptr(w[33 * C:]), _lib.ptr(ls), _lib.ptr(lb), n_pix, C, ACT[act], 1, 1e-5, 0.1, phase, 0, None, 0, st())

        def bwd(i, phase):
            w, r = ws[i], red[i]
            return lib.dfine_bn_act_bwd(_lib.ptr(x[i]), None, _lib.ptr(dy[i]), 0, None, 0, 1, _lib.ptr(w[32 * C:]), _lib.ptr(w[33 * C:]), _lib.ptr(gamma), _lib.ptr(beta),
                                        _lib.ptr(ls), _lib.ptr(r), _lib.ptr(y[i]), None, _lib.ptr(r[64 * C:]), _lib.ptr(r[65 * C:]),
                                        _lib.ptr(r[66 * C:]) if lab else None, _lib.ptr(r[66 * C + 1:]) if lab else None, n_pix, C, ACT[act], 1, 1e-5, phase, st())

        for i in range(n_sets):
            assert fwd(i, 1) == 0 and fwd(i, 2) == 0
        t_copy = timed_graph(lambda i: y[i].copy_(x[i]), n_sets, reps)
        t_stats = timed_graph(lambda i: fwd(i, 1), n_sets, reps)
        t_apply = timed_graph(lambda i: fwd(i, 2), n_sets, reps)
        t_red = timed_graph(lambda i: bwd(i, 1), n_sets, reps)
        t_dx = timed_graph(lambda i: bwd(i, 2), n_sets, reps)
        tb = lambda t, passes: passes * S / t / 1e6
        print(f"{n_pix:8d}x{C:<4d} {act:5s} S={S / 1e6:6.1f} MB | copy {t_copy:7.1f} us {tb(t_copy, 2):5.2f} TB/s | stats {t_stats:7.1f} {tb(t_stats, 1):5.2f} | apply {t_apply:7.1f} {tb(t_apply, 2):5.2f} | "
              f"reduce {t_red:7.1f} {tb(t_red, 2):5.2f} | dx {t_dx:7.1f} {tb(t_dx, 3):5.2f}", flush=True)
        del x, dy, y


if "--tail" in sys.argv:  # where does the backward reduction's time go: full, without its global atomics, without any column reduction
    for dbg in ("0", "1", "2"):
        os.environ["DFINE_BN_DEBUG"] = dbg
        run(f"DFINE_BN_DEBUG={dbg}")
elif "--sweep" in sys.argv:
    for rb, un in itertools.product((512, 1024, 2048), (2, 4, 8)):
        os.environ["DFINE_BN_REDUCE_BLOCKS"], os.environ["DFINE_BN_BWD_UNROLL"] = str(rb), str(un)
        run(f"reduce blocks {rb}, bwd unroll {un}")
else:
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        os.environ[k] = v
    run(" ".join(sys.argv[1:]) or "defaults")
