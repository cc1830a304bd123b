"""Grouped (deferred) weight gradients: dfine_wgrad_group_* (include/dfine_hip.h). The planner is host code (CPU tests);
the launch is compared with the single-problem kernels and a torch fp32 reference on the GPU."""
import ctypes
import struct

import numpy as np
import pytest
import torch


def _problem(_lib, conv, **kw):
    p = _lib.WgradProblem(conv=int(conv), **kw)
    return p


def _fake(addr):
    return ctypes.c_void_p(addr)


def _plan(lib, _lib, problems):
    n = len(problems)
    arr = (_lib.WgradProblem * n)(*problems)
    need = lib.dfine_wgrad_group_plan(arr, n, None, 0)
    assert need > 0, need
    buf = (ctypes.c_char * need)()
    used = lib.dfine_wgrad_group_plan(arr, n, buf, need)
    assert used == need
    return bytes(buf)


def _parse(table):
    """-> (problems [(M, N, K, m_chunk, k_tiles, variant, n_chunks)], launches [(variant, blocks [(p, chunk, tile)])])"""
    magic, n_problems, n_launches, _, problems_off, total = struct.unpack_from("<iiiiqq", table, 0)
    assert magic == 0x57475250 and total == len(table)
    launches = []
    for l in range(n_launches):
        variant, grid, off = struct.unpack_from("<iiq", table, 32 + 16 * l)
        blocks = np.frombuffer(table, dtype=np.int32, count=4 * grid, offset=off).reshape(grid, 4)
        launches.append((variant, blocks))
    problems = []
    for i in range(n_problems):
        base = problems_off + 112 * i
        M, N, K, ldw, m_chunk, k_tiles, variant, n_chunks = struct.unpack_from("<8i", table, base + 32)
        problems.append((M, N, K, m_chunk, k_tiles, variant, n_chunks))
    return problems, launches


def test_plan_covers_every_tile_of_every_chunk_exactly_once(hip_lib):
    """No GPU: the block tables list each (problem, row chunk, output tile) once, chunks cover all rows, the tiles of one
    chunk sit on one XCD (ids of one residue class mod 8) and padding entries are marked."""
    from dfine_seg_amd import _lib

    A = 1 << 20  # fake, 16-byte aligned addresses: the planner never dereferences them
    problems = [
        _problem(_lib, 0, dY=A, X=A, dW=A, db=A, M=15744, N=256, K=256, ldw=256, ldy=256, ldx=256),
        _problem(_lib, 0, dY=A, X=A, dW=A, db=None, M=15744, N=80, K=256, ldw=256, ldy=80, ldx=256),
        _problem(_lib, 0, dY=A, X=A, dW=A, db=None, M=100, N=1024, K=256, ldw=256, ldy=1024, ldx=256),
        _problem(_lib, 1, dY=A, X=A, dW=A, db=None, B=32, H=40, W=40, CI=128, OH=40, OW=40, CO=128, KH=3, KW=3, stride=1, pad=1, ldy=128, ldx=128),
        _problem(_lib, 1, dY=A, X=A, dW=A, db=None, B=32, H=80, W=80, CI=64, OH=40, OW=40, CO=96, KH=3, KW=3, stride=2, pad=1, ldy=96, ldx=64),
        _problem(_lib, 1, dY=A, X=A, dW=A, db=None, B=32, H=40, W=40, CI=384, OH=40, OW=40, CO=768, KH=1, KW=1, stride=1, pad=0, ldy=768, ldx=384),
    ]
    parsed, launches = _parse(_plan(hip_lib, _lib, problems))
    assert len(parsed) == len(problems)
    seen = {}
    for variant, blocks in launches:
        tn, tk = (128 if variant & 2 and variant < 8 else 64), (128 if variant & 1 and variant < 8 else 64)
        assert len(blocks) % 8 == 0
        for bid, (p, chunk, tile, _) in enumerate(blocks):
            if p < 0:
                continue
            M, N, K, m_chunk, k_tiles, pv, n_chunks = parsed[p]
            assert pv == variant
            tiles = -(-N // tn) * -(-K // tk)
            assert k_tiles == -(-K // tk) and 0 <= tile < tiles and 0 <= chunk < n_chunks
            assert (p, chunk, tile) not in seen
            seen[(p, chunk, tile)] = bid % 8
    for p, (M, N, K, m_chunk, k_tiles, variant, n_chunks) in enumerate(parsed):
        tn, tk = (128 if variant & 2 and variant < 8 else 64), (128 if variant & 1 and variant < 8 else 64)
        tiles = -(-N // tn) * -(-K // tk)
        assert m_chunk % 128 == 0 and (n_chunks - 1) * m_chunk < M <= n_chunks * m_chunk
        for c in range(n_chunks):
            xcds = {seen[(p, c, t)] for t in range(tiles)}  # KeyError: a tile is missing
            assert len(xcds) == 1
    # the 1x1 convolution is planned as a plain GEMM (variant without the gather bit), the 3x3 ones are gathered
    assert parsed[5][5] & 4 == 0 and parsed[3][5] & 4 and parsed[4][5] & 4


def test_ineligible_problems_are_refused(hip_lib):
    from dfine_seg_amd import _lib

    A = 1 << 20
    ok = _problem(_lib, 0, dY=A, X=A, dW=A, db=None, M=64, N=64, K=64, ldw=64, ldy=64, ldx=64)
    assert hip_lib.dfine_wgrad_group_eligible(ctypes.byref(ok)) == 1
    # Linear problems that are not 16-byte friendly are served by the narrow-load variant (the 132- / 4-wide heads, K = 20)
    for narrow in (
        _problem(_lib, 0, dY=A, X=A, dW=A, db=None, M=64, N=132, K=64, ldw=64, ldy=132, ldx=64),
        _problem(_lib, 0, dY=A + 2, X=A, dW=A, db=None, M=64, N=64, K=64, ldw=64, ldy=64, ldx=64),
        _problem(_lib, 0, dY=A, X=A, dW=A, db=A, M=64, N=64, K=20, ldw=20, ldy=64, ldx=20),
    ):
        assert hip_lib.dfine_wgrad_group_eligible(ctypes.byref(narrow)) == 1
        parsed, launches = _parse(_plan(hip_lib, _lib, [ok, narrow]))
        assert parsed[1][5] == 8 and {v for v, _ in launches} == {0, 8}
    for bad in (
        _problem(_lib, 0, dY=A + 1, X=A, dW=A, db=None, M=64, N=64, K=64, ldw=64, ldy=64, ldx=64),  # not even bf16-aligned
        _problem(_lib, 1, dY=A, X=A, dW=A, db=None, B=2, H=8, W=8, CI=3, OH=4, OW=4, CO=24, KH=3, KW=3, stride=2, pad=1, ldy=24, ldx=3),
        _problem(_lib, 0, dY=A, X=A, dW=None, db=None, M=64, N=64, K=64, ldw=64, ldy=64, ldx=64),
        _problem(_lib, 0, dY=A, X=A, dW=A, db=None, M=64, N=64, K=64, ldw=32, ldy=64, ldx=64),      # ldw < K
    ):
        assert hip_lib.dfine_wgrad_group_eligible(ctypes.byref(bad)) == 0
        arr = (_lib.WgradProblem * 2)(ok, bad)
        assert hip_lib.dfine_wgrad_group_plan(arr, 2, None, 0) == -2


@pytest.mark.gpu
def test_grouped_launch_equals_single_launches_and_fp32_reference(hip_lib):
    """A mixed bag of Linear and conv problems (all tile shapes, strides, channel-slice pitches, a bias) through ONE grouped
    launch: equal to the single-problem kernels up to the order of fp32 atomics (1e-5 relative to the tensor's scale), and
    to the fp32 autograd reference within bf16-product round-off."""
    from dfine_seg_amd import _lib, ops

    torch.manual_seed(0)
    dev = "cuda"
    keep, problems, singles, refs = [], [], [], []

    def lin(M, N, K, bias, pitch=0):
        dy_full = torch.randn(M, N + pitch, device=dev).bfloat16()
        x_full = torch.randn(M, K + pitch, device=dev).bfloat16()
        dy, x = dy_full[:, :N], x_full[:, :K]
        dw, db = torch.zeros(N, K, device=dev), (torch.zeros(N, device=dev) if bias else None)
        dw1, db1 = torch.zeros_like(dw), (torch.zeros_like(db) if bias else None)
        keep.append((dy_full, x_full))
        problems.append(_lib.WgradProblem(dY=dy.data_ptr(), X=x.data_ptr(), dW=dw.data_ptr(), db=db.data_ptr() if bias else None, conv=0,
                                          M=M, N=N, K=K, ldw=K, ldy=N + pitch, ldx=K + pitch))
        rc = hip_lib.dfine_linear_wgrad(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw1), _lib.ptr(db1), M, N, K, K, N + pitch, K + pitch, _lib.stream_ptr())
        assert rc == 0
        ref_w = dy.float().t() @ x.float()
        singles.append((dw, dw1, ref_w))
        if bias:
            singles.append((db, db1, dy.float().sum(0)))

    def conv(B, H, ci, co, k, stride, pad):
        x = torch.randn(B, H, H, ci, device=dev).bfloat16()
        OH = (H + 2 * pad - k) // stride + 1
        dy = torch.randn(B, OH, OH, co, device=dev).bfloat16()
        dw, dw1 = torch.zeros(co, k, k, ci, device=dev), torch.zeros(co, k, k, ci, device=dev)
        keep.append((x, dy))
        problems.append(_lib.WgradProblem(dY=dy.data_ptr(), X=x.data_ptr(), dW=dw.data_ptr(), db=None, conv=1, B=B, H=H, W=H, CI=ci, OH=OH,
                                          OW=OH, CO=co, KH=k, KW=k, stride=stride, pad=pad, ldy=co, ldx=ci))
        rc = hip_lib.dfine_conv_wgrad(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw1), B, H, H, ci, OH, OH, co, k, k, stride, pad, co, ci, _lib.stream_ptr())
        assert rc == 0
        xr = x.float().permute(0, 3, 1, 2).requires_grad_(False)
        w = torch.zeros(co, ci, k, k, device=dev, requires_grad=True)
        y = torch.nn.functional.conv2d(xr, w, stride=stride, padding=pad)
        (gw,) = torch.autograd.grad(y, w, dy.float().permute(0, 3, 1, 2))
        singles.append((dw, dw1, gw.permute(0, 2, 3, 1)))

    lin(3000, 256, 256, True)
    lin(3000, 80, 256, False)
    lin(777, 1024, 64, True)          # ragged M (not a multiple of the 32-row step)
    lin(5000, 64, 1024, False, pitch=8)  # channel slices of wider rows
    lin(40, 24, 48, False)             # a single short chunk
    lin(3000, 132, 256, True)          # narrow-load variant: N % 8 != 0
    lin(900, 4, 256, True)
    lin(900, 64, 20, False)            # K % 8 != 0
    conv(4, 40, 128, 128, 3, 1, 1)
    conv(4, 40, 64, 96, 3, 2, 1)
    conv(4, 20, 384, 768, 1, 1, 0)
    conv(2, 40, 32, 32, 3, 1, 1)
    conv(2, 24, 48, 24, 3, 2, 1)
    n = len(problems)
    arr = (_lib.WgradProblem * n)(*problems)
    need = hip_lib.dfine_wgrad_group_plan(arr, n, None, 0)
    assert need > 0
    host = torch.empty(need, dtype=torch.uint8)
    assert hip_lib.dfine_wgrad_group_plan(arr, n, ctypes.c_void_p(host.data_ptr()), need) == need
    table = host.to(dev)
    assert hip_lib.dfine_wgrad_group_launch(ctypes.c_void_p(host.data_ptr()), _lib.ptr(table), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    for got, single, ref in singles:
        scale = float(ref.abs().max())
        assert float((got - single).abs().max()) <= 1e-5 * scale + 1e-6
        assert float((got - ref).abs().max()) <= 2e-3 * scale  # products of bf16 values are exact in fp32; only the sum order differs


@pytest.mark.gpu
def test_queue_defers_until_flush_and_matches_immediate_gradients(hip_lib):
    """ops.WgradQueue through autograd: with a queue installed Linear / conv backward leaves .grad untouched until flush();
    afterwards it equals what the immediate path accumulates."""
    from dfine_seg_amd import ops

    torch.manual_seed(1)
    lin = torch.nn.Linear(256, 192).cuda()
    conv = torch.nn.Conv2d(64, 96, 3, padding=1, bias=False).cuda().to(memory_format=torch.channels_last)
    x1 = torch.randn(4, 300, 256, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    x2 = torch.randn(4, 64, 20, 20, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)

    def run(queue):
        for p in list(lin.parameters()) + list(conv.parameters()):
            p.grad = torch.zeros_like(p)
        ops.wgrad_queue = queue
        try:
            y = ops.linear(x1, lin.weight, lin.bias).float().square().sum() + ops.dense_conv(x2, conv.weight, 1, 1).float().square().sum()
            y.backward()
            if queue is not None:
                assert float(lin.weight.grad.abs().max()) == 0.0 and float(conv.weight.grad.abs().max()) == 0.0
                assert len(queue.problems) == 2
                ops.join_wgrad()
                assert len(queue.problems) == 0 and queue.launches == 1
        finally:
            ops.wgrad_queue = None
        return [p.grad.clone() for p in list(lin.parameters()) + list(conv.parameters())]

    now = run(None)
    later = run(ops.WgradQueue())
    for a, b in zip(now, later):
        assert float(a.abs().max()) > 0
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
