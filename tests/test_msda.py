"""Deformable-attention core: oracle vs the reference's golden outputs (CPU), and the HIP
kernels vs golden / oracle (GPU, through the C ABI)."""
import pytest
import torch

from conftest import load_golden
from oracle import ops as oracle_ops

CASES = ["small", "rect3"]


def _case(g, name):
    t = {k.split(".", 1)[1]: torch.from_numpy(v) for k, v in g.items() if k.startswith(name + ".")}
    t["shapes"] = [tuple(int(x) for x in s) for s in t["shapes"]]
    t["num_points"] = [int(x) for x in t["num_points"]]
    return t


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden(name):
    c = _case(load_golden("msda_core"), name)
    value, loc, attn = (c[k].clone().requires_grad_(True) for k in ("value", "loc", "attn"))
    out = oracle_ops.deformable_attention_core(value, c["shapes"], loc, attn, c["num_points"])
    out.backward(c["grad_out"])
    # fp32 CPU restatement vs the reference's own output: reassociation-level differences only
    torch.testing.assert_close(out, c["out"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(value.grad, c["grad_value"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(loc.grad, c["grad_loc"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(attn.grad, c["grad_attn"], rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------ GPU
def _hip_fwd_bwd(value, shapes, loc, attn, npts, grad_out):
    from dfine_seg_amd import ops

    value = value.detach().cuda().requires_grad_(True)
    loc = loc.detach().cuda().requires_grad_(True)
    attn = attn.detach().cuda().requires_grad_(True)
    out = ops.deformable_attention_core(value, shapes, loc, attn, npts)
    out.backward(grad_out.cuda().to(out.dtype))
    return out.float().cpu(), value.grad.float().cpu(), loc.grad.cpu(), attn.grad.cpu()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_f32_matches_reference_golden(hip_lib, name):
    c = _case(load_golden("msda_core"), name)
    out, gv, gl, ga = _hip_fwd_bwd(c["value"], c["shapes"], c["loc"], c["attn"], c["num_points"], c["grad_out"])
    # fp32 kernel, fp32 accumulate: tolerance = summation-order noise (atomics are unordered)
    torch.testing.assert_close(out, c["out"], rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(gv, c["grad_value"], rtol=1e-5, atol=2e-5)
    torch.testing.assert_close(gl, c["grad_loc"], rtol=1e-4, atol=2e-4)
    torch.testing.assert_close(ga, c["grad_attn"], rtol=1e-5, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_bf16_matches_reference_golden(hip_lib, name):
    """bf16 value/out storage, fp32 math: the only error is the bf16 rounding of value
    (rel 2^-9) going in and of out / grad_out coming out."""
    c = _case(load_golden("msda_core"), name)
    out, gv, gl, ga = _hip_fwd_bwd(
        c["value"].bfloat16(), c["shapes"], c["loc"], c["attn"], c["num_points"], c["grad_out"]
    )
    torch.testing.assert_close(out, c["out"], rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(gv, c["grad_value"], rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(ga, c["grad_attn"], rtol=2e-2, atol=6e-2)
    # grad_loc is a difference of neighbouring bf16-rounded pixels times the map size
    rel = (gl - c["grad_loc"]).norm() / c["grad_loc"].norm()
    assert rel < 2e-2, rel


def _random_case(B, Lq, nH, c, shapes, npts, seed, spread=1.3):
    g = torch.Generator().manual_seed(seed)
    S, P = sum(h * w for h, w in shapes), sum(npts)
    value = torch.randn(B, S, nH, c, generator=g)
    loc = torch.rand(B, Lq, nH, P, 2, generator=g) * spread - (spread - 1) / 2
    attn = torch.softmax(torch.randn(B, Lq, nH, P, generator=g), -1)
    grad_out = torch.randn(B, Lq, nH * c, generator=g)
    return value, loc, attn, grad_out


@pytest.mark.gpu
@pytest.mark.parametrize(
    "B,Lq,nH,c,shapes,npts",
    [
        (2, 492, 8, 32, [(80, 80), (40, 40), (20, 20)], [3, 6, 3]),  # D-FINE-m/s @640 (BASELINE configs 1-3)
        (2, 492, 8, 16, [(20, 20), (10, 10)], [6, 6]),  # D-FINE-n @320 (config 0)
        (1, 300, 8, 32, [(120, 120), (60, 60), (30, 30)], [3, 6, 3]),  # D-FINE-x @960 eval queries
        (3, 7, 1, 32, [(1, 1)], [1]),  # degenerate: one pixel, one point, ragged thread tail
        (1, 33, 8, 32, [(5, 7), (3, 2)], [4, 4]),
    ],
)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_hip_matches_oracle_config_shapes(hip_lib, B, Lq, nH, c, shapes, npts, dtype):
    value, loc, attn, grad_out = _random_case(B, Lq, nH, c, shapes, npts, seed=7)
    value = value.to(dtype)
    grad_out = grad_out.to(dtype)
    v = value.float().clone().requires_grad_(True)
    l, a = loc.clone().requires_grad_(True), attn.clone().requires_grad_(True)
    ref = oracle_ops.deformable_attention_core(v, shapes, l, a, npts)
    ref.backward(grad_out.float())
    out, gv, gl, ga = _hip_fwd_bwd(value, shapes, loc, attn, npts, grad_out)
    if dtype == torch.float32:
        tol = dict(rtol=1e-4, atol=1e-4)
    else:  # bf16 rounding of out (fwd) and of the value-dtype grad_value (bwd)
        tol = dict(rtol=1.6e-2, atol=1.6e-2)
    torch.testing.assert_close(out, ref.detach(), **tol)
    torch.testing.assert_close(ga, a.grad, rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(gl, l.grad, rtol=1e-3, atol=1e-2 * max(max(s) for s in shapes) / 20)
    gv_tol = tol if dtype == torch.float32 else dict(rtol=1.6e-2, atol=3e-2)
    torch.testing.assert_close(gv, v.grad, **gv_tol)


@pytest.mark.gpu
def test_hip_properties_full_size(hip_lib):
    """BASELINE full size (B=32 m@640): properties that need no oracle run.
    (1) locations wholly outside the image gather zeros; (2) linearity in the attention
    weights; (3) a constant map samples to sum(attn) wherever all 4 corners are inside."""
    from dfine_seg_amd import ops

    B, Lq, nH, c, shapes, npts = 32, 492, 8, 32, [(80, 80), (40, 40), (20, 20)], [3, 6, 3]
    value, loc, attn, _ = _random_case(B, Lq, nH, c, shapes, npts, seed=11)
    value, loc, attn = value.cuda().bfloat16(), loc.cuda(), attn.cuda()
    out1 = ops.deformable_attention_core(value, shapes, loc, attn, npts).float()
    out2 = ops.deformable_attention_core(value, shapes, loc, 2 * attn, npts).float()
    torch.testing.assert_close(out2, 2 * out1, rtol=1e-2, atol=1e-2)
    far = ops.deformable_attention_core(value, shapes, loc + 3.0, attn, npts)
    assert float(far.float().abs().max()) == 0.0
    ones = torch.ones_like(value)
    inner = loc.clamp(0.1, 0.9)
    const = ops.deformable_attention_core(ones, shapes, inner, attn, npts).float()
    torch.testing.assert_close(const, attn.sum(-1, keepdim=True).expand(B, Lq, nH, c).reshape(B, Lq, nH * c), rtol=1e-2, atol=1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("c,shapes,npts", [(32, [(80, 80), (40, 40), (20, 20)], [3, 6, 3]), (16, [(20, 20), (10, 10)], [6, 6]),
                                           (32, [(37, 23), (9, 14)], [2, 5])])
def test_value_gradient_without_global_atomics_matches_the_atomic_kernel(hip_lib, c, shapes, npts):
    """dfine_msda_bwd_points + dfine_msda_bwd_value (LDS slabs, several calls batched, every element written once) against the
    float-atomic backward: same grad_loc / grad_attn and value gradient to fp32 round-off (the order of the adds
    differs), and -- no global atomics involved -- identical bits from run to run."""
    import ctypes

    from dfine_seg_amd import _lib

    B, Lq, nH = 3, 77, 8
    S, P = sum(h * w for h, w in shapes), sum(npts)
    g = torch.Generator(device="cuda").manual_seed(c + S)
    value = torch.randn(B, S, nH, c, device="cuda", generator=g).bfloat16()
    calls = []
    for _ in range(3):
        loc = (torch.rand(B, Lq, nH, P, 2, device="cuda", generator=g) * 1.2 - 0.1).contiguous()
        attn = torch.softmax(torch.randn(B, Lq, nH, P, device="cuda", generator=g), -1).contiguous()
        go = torch.randn(B, Lq, nH * c, device="cuda", generator=g).bfloat16()
        calls.append((loc, attn, go))
    hw = _lib.i32_array(v for pair in shapes for v in pair)
    np_ = _lib.i32_array(npts)
    ref_value = torch.zeros(B, S, nH, c, device="cuda")
    ref_pts = []
    for loc, attn, go in calls:
        gl, ga = torch.empty_like(loc), torch.empty_like(attn)
        assert hip_lib.dfine_msda_bwd(_lib.ptr(value), 1, _lib.ptr(loc), _lib.ptr(attn), _lib.ptr(go), _lib.ptr(ref_value), 0, _lib.ptr(gl),
                                      _lib.ptr(ga), B, S, Lq, nH, c, len(npts), hw, np_, None) == 0
        ref_pts.append((gl, ga))
    outs = []
    for _ in range(2):
        for (loc, attn, go), (gl_ref, ga_ref) in zip(calls, ref_pts):
            gl, ga = torch.empty_like(loc), torch.empty_like(attn)
            assert hip_lib.dfine_msda_bwd_points(_lib.ptr(value), 1, _lib.ptr(loc), _lib.ptr(attn), _lib.ptr(go), _lib.ptr(gl), _lib.ptr(ga),
                                                 B, S, Lq, nH, c, len(npts), hw, np_, None) == 0
            # (dfine_msda_bwd_points sums a head's channels 8 per lane and then over 4 lanes, the scatter kernel over 32 lanes)
            torch.testing.assert_close(gl, gl_ref, rtol=1e-4, atol=1e-5 * float(gl_ref.abs().max()))
            torch.testing.assert_close(ga, ga_ref, rtol=1e-4, atol=1e-5 * float(ga_ref.abs().max()))
        out = torch.full((B, S, nH, c), float("nan"), device="cuda")  # every element must be written
        arr = lambda k: (ctypes.c_void_p * len(calls))(*[t[k].data_ptr() for t in calls])
        assert hip_lib.dfine_msda_bwd_value(1, arr(0), arr(1), arr(2), len(calls), _lib.ptr(out), B, S, Lq, nH, c, len(npts), hw, np_, None) == 0
        outs.append(out)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0]).all()
    torch.testing.assert_close(outs[0], ref_value, rtol=1e-4, atol=1e-4)
    # (LDS float atomics still reorder adds inside a slab, so the two runs agree to round-off, not bit for bit)
    torch.testing.assert_close(outs[0], outs[1], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("c,shapes,npts", [(32, [(80, 80), (40, 40), (20, 20)], [3, 6, 3]), (16, [(20, 20), (10, 10)], [6, 6]),
                                           (32, [(37, 23), (9, 14)], [2, 5])])
@pytest.mark.parametrize("piled,B", [(False, 3), (True, 3), (False, 33)])
def test_sorted_value_gradient_matches_the_atomic_kernel(hip_lib, c, shapes, npts, piled, B):
    """dfine_msda_bwd_value_sorted (each (image, head)'s hits counting-sorted by token in LDS, one half-wave gathers each list;
    no float atomics, every element written once) against the float-atomic scatter over the same calls: equal to fp32
    round-off (1e-4: the order of the adds differs). ``piled``: all queries of an image sample around ONE point -- long lists
    on a few tokens, empty ones everywhere else (what matched queries do to the tokens of an object)."""
    import ctypes

    from dfine_seg_amd import _lib

    Lq, nH = 77, 8  # B = 3: 24 (image, head) pairs, token ranges split over 16 workgroups each; B = 33: one workgroup per pair
    S, P = sum(h * w for h, w in shapes), sum(npts)
    g = torch.Generator(device="cuda").manual_seed(c + S + int(piled))
    value = torch.randn(B, S, nH, c, device="cuda", generator=g).bfloat16()
    calls = []
    for _ in range(4):
        if piled:
            centre = torch.rand(B, 1, 1, 1, 2, device="cuda", generator=g) * 0.6 + 0.2
            loc = (centre + torch.randn(B, Lq, nH, P, 2, device="cuda", generator=g) * 0.01).contiguous()
        else:
            loc = (torch.rand(B, Lq, nH, P, 2, device="cuda", generator=g) * 1.2 - 0.1).contiguous()
        attn = torch.softmax(torch.randn(B, Lq, nH, P, device="cuda", generator=g), -1).contiguous()
        go = torch.randn(B, Lq, nH * c, device="cuda", generator=g).bfloat16()
        calls.append((loc, attn, go))
    hw = _lib.i32_array(v for pair in shapes for v in pair)
    np_ = _lib.i32_array(npts)
    ref_value = torch.zeros(B, S, nH, c, device="cuda")
    for loc, attn, go in calls:
        gl, ga = torch.empty_like(loc), torch.empty_like(attn)
        assert hip_lib.dfine_msda_bwd(_lib.ptr(value), 1, _lib.ptr(loc), _lib.ptr(attn), _lib.ptr(go), _lib.ptr(ref_value), 0, _lib.ptr(gl),
                                      _lib.ptr(ga), B, S, Lq, nH, c, len(npts), hw, np_, None) == 0
    cap = hip_lib.dfine_msda_sorted_records(len(calls), B, S, Lq, nH, P)
    assert cap >= len(calls) * B * Lq * nH * P * 4
    entries = torch.full((cap, 2), -7, dtype=torch.int32, device="cuda")  # contents irrelevant on entry
    arr = lambda k: (ctypes.c_void_p * len(calls))(*[t[k].data_ptr() for t in calls])
    for _ in range(2):
        out = torch.full((B, S, nH, c), float("nan"), device="cuda")  # every element must be written
        assert hip_lib.dfine_msda_bwd_value_sorted(1, arr(0), arr(1), arr(2), len(calls), _lib.ptr(out), 0, _lib.ptr(entries), cap, B, S, Lq, nH, c,
                                                   len(npts), hw, np_, None) == 0
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        scale = float(ref_value.abs().max())
        torch.testing.assert_close(out, ref_value, rtol=1e-4, atol=1e-5 * scale)
    # bf16 output (what the training step asks for: the memory's dtype): the same fp32 sums, rounded once
    out16 = torch.full((B, S, nH, c), float("nan"), device="cuda", dtype=torch.bfloat16)
    assert hip_lib.dfine_msda_bwd_value_sorted(1, arr(0), arr(1), arr(2), len(calls), _lib.ptr(out16), 1, _lib.ptr(entries), cap, B, S, Lq, nH, c,
                                               len(npts), hw, np_, None) == 0
    torch.cuda.synchronize()
    torch.testing.assert_close(out16.float(), out, rtol=2 ** -7, atol=1e-5 * scale)  # one bf16 rounding of (nearly) the same sum
    # too small a record buffer is refused, not overrun
    assert hip_lib.dfine_msda_bwd_value_sorted(1, arr(0), arr(1), arr(2), len(calls), _lib.ptr(out), 0, _lib.ptr(entries), cap - 1, B, S, Lq, nH, c,
                                               len(npts), hw, np_, None) == -1


@pytest.mark.gpu
@pytest.mark.parametrize("c,shapes,npts", [(32, [(80, 80), (40, 40), (20, 20)], [3, 6, 3]), (16, [(20, 20), (10, 10)], [6, 6])])
def test_packed_bf16_value_gradient_matches_the_float_one(hip_lib, c, shapes, npts):
    """dfine_msda_bwd with a bf16 gradient map (global_atomic_pk_add_bf16, two channels per add) against the float-atomic
    form on the same inputs, four calls accumulating into one map like the four decoder layers: grad_loc / grad_attn bit-equal;
    the value gradient within bf16 ACCUMULATION round-off -- every add rounds the running sum to 8 mantissa bits, so with n
    adds per element the error random-walks to ~sqrt(n) * 2^-9 of the sum: relative L2 < 1.5e-2 over the map, and no element
    off by more than 6 % of the map's largest magnitude."""
    from dfine_seg_amd import _lib

    B, Lq, nH = 4, 300, 8
    S, P = sum(h * w for h, w in shapes), sum(npts)
    g = torch.Generator(device="cuda").manual_seed(7 + c)
    value = torch.randn(B, S, nH, c, device="cuda", generator=g).bfloat16()
    hw = _lib.i32_array(v for pair in shapes for v in pair)
    np_ = _lib.i32_array(npts)
    g32 = torch.zeros(B, S, nH, c, device="cuda")
    g16 = torch.zeros(B, S, nH, c, device="cuda", dtype=torch.bfloat16)
    for _ in range(4):
        loc = (torch.rand(B, Lq, nH, P, 2, device="cuda", generator=g) * 1.2 - 0.1).contiguous()
        attn = torch.softmax(torch.randn(B, Lq, nH, P, device="cuda", generator=g), -1).contiguous()
        go = torch.randn(B, Lq, nH * c, device="cuda", generator=g).bfloat16()
        gl0, ga0, gl1, ga1 = torch.empty_like(loc), torch.empty_like(attn), torch.empty_like(loc), torch.empty_like(attn)
        assert hip_lib.dfine_msda_bwd(_lib.ptr(value), 1, _lib.ptr(loc), _lib.ptr(attn), _lib.ptr(go), _lib.ptr(g32), 0, _lib.ptr(gl0),
                                      _lib.ptr(ga0), B, S, Lq, nH, c, len(npts), hw, np_, None) == 0
        assert hip_lib.dfine_msda_bwd(_lib.ptr(value), 1, _lib.ptr(loc), _lib.ptr(attn), _lib.ptr(go), _lib.ptr(g16), 1, _lib.ptr(gl1),
                                      _lib.ptr(ga1), B, S, Lq, nH, c, len(npts), hw, np_, None) == 0
        assert torch.equal(gl0, gl1) and torch.equal(ga0, ga1)
    torch.cuda.synchronize()
    diff = (g16.float() - g32)
    assert float(diff.norm() / g32.norm()) < 1.5e-2, float(diff.norm() / g32.norm())
    assert float(diff.abs().max()) < 0.06 * float(g32.abs().max())
    # fp32 values with a bf16 map are refused (the packed form is for the bf16 training path)
    v32 = value.float()
    assert hip_lib.dfine_msda_bwd(_lib.ptr(v32), 0, _lib.ptr(loc), _lib.ptr(attn), _lib.ptr(go.float()), _lib.ptr(g16), 1, _lib.ptr(gl1),
                                  _lib.ptr(ga1), B, S, Lq, nH, c, len(npts), hw, np_, None) == -2


@pytest.mark.gpu
@pytest.mark.parametrize("pts", [[3, 6, 3], [2, 3]])  # 12 points: the register path; 5: the element-by-element one
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_prepare_matches_the_op_by_op_form(dtype, pts):
    """Locations and softmaxed weights from the two projections (dfine_decoder.py:148-166), forward and backward, against
    the same arithmetic spelled with ATen ops on the same inputs. fp32 intermediate math both ways: 1e-6 on fp32 inputs;
    bf16 inputs differ only in the final rounding of the two input gradients (one bf16 ulp)."""
    from dfine_seg_amd import ops

    torch.manual_seed(3)
    dev = "cuda"
    B, Lq, nH = 3, 77, 8
    P = sum(pts)
    scale = torch.tensor([1.0 / n for n in pts for _ in range(n)], device=dev)
    off = torch.randn(B, Lq, nH * P * 2, device=dev).to(dtype)
    lg = (2 * torch.randn(B, Lq, nH * P, device=dev)).to(dtype)
    ref = torch.rand(B, Lq, 4, device=dev) * 0.5 + 0.1
    g_loc = torch.randn(B, Lq, nH, P, 2, device=dev)
    g_w = torch.randn(B, Lq, nH, P, device=dev)

    def run(fused):
        o, l = off.clone().requires_grad_(True), lg.clone().requires_grad_(True)
        if fused:
            loc, w = ops.msda_prepare(o, l, ref, scale, nH)
        else:
            of = o.float().view(B, Lq, nH, P, 2)
            w = torch.softmax(l.float().view(B, Lq, nH, P), dim=-1)
            r = ref[:, :, None, None, :]
            loc = r[..., :2] + of * (scale.view(P, 1) * 0.5) * r[..., 2:]
        ((loc * g_loc).sum() + (w * g_w).sum()).backward()
        return loc.detach(), w.detach(), o.grad, l.grad

    a, b = run(True), run(False)
    assert a[0].dtype == torch.float32 and a[1].dtype == torch.float32 and a[2].dtype == dtype and a[3].dtype == dtype
    torch.testing.assert_close(a[0], b[0], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(a[1], b[1], rtol=1e-5, atol=1e-7)
    tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-3)
    torch.testing.assert_close(a[2].float(), b[2].float(), **tol)
    torch.testing.assert_close(a[3].float(), b[3].float(), **tol)
