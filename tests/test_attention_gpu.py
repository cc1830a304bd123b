"""Self-attention kernels (csrc/attention.hip) through the C ABI vs a plain PyTorch fp32 restatement of
softmax(q k^T / sqrt(dh) + mask) v on the same bf16-rounded operands: forward, and dq / dk / dv against fp32 autograd.
Shapes: decoder self-attention (L = 492 with the denoising mask, head_dim 32; D-FINE-n: head_dim 16), AIFI (L = 400 / 900,
head_dim 32 / 48, no mask), ragged L (not a multiple of the 64-row tiles), a fully blocked row.
Tolerance: P and dS are rounded to bf16 before the second product (like every flash-style kernel) and the outputs are
bf16; dS = P * (dP - D) cancels, so single elements can be off by a few bf16 ulps of the LARGEST term: the bound is on
the relative L2 error of the whole tensor (< 1e-2) plus a loose element-wise cap (|err| < 0.1 * rms(ref))."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(qk, v, blocked, H):
    B, L, D2 = qk.shape
    D = D2 // 2
    dh = D // H
    q, k = qk[..., :D].view(B, L, H, dh).transpose(1, 2), qk[..., D:].view(B, L, H, dh).transpose(1, 2)
    vv = v.view(B, L, H, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / dh ** 0.5
    if blocked is not None:
        s = s.masked_fill(blocked, float("-inf"))
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p, nan=0.0)  # a fully blocked row attends to nothing
    return (p @ vv).transpose(1, 2).reshape(B, L, D)


def _close(got, ref, what):
    got, ref = got.float(), ref.float()
    rms = ref.pow(2).mean().sqrt().item() + 1e-6
    rel_l2 = ((got - ref).pow(2).sum().sqrt() / (ref.pow(2).sum().sqrt() + 1e-12)).item()
    worst = (got - ref).abs().max().item()
    assert rel_l2 < 1e-2 and worst < 0.1 * rms, f"{what}: rel-L2 {rel_l2:.3e}, max |err| {worst:.3e} vs rms {rms:.3e}"


def _dn_mask(L, Q=300, groups=3):
    nd = L - Q
    m = torch.zeros(L, L, dtype=torch.bool)
    if nd > 0:
        per = nd // groups
        m[nd:, :nd] = True
        for g in range(groups):
            lo, hi = per * g, per * (g + 1) if g + 1 < groups else nd
            m[lo:hi, :lo] = True
            m[lo:hi, hi:nd] = True
    return m


@pytest.mark.parametrize("B,L,H,dh,masked", [(3, 492, 8, 32, True), (2, 400, 8, 32, False), (2, 900, 8, 48, False),
                                             (2, 330, 8, 16, True), (1, 77, 4, 32, True), (2, 100, 8, 16, False), (1, 64, 2, 64, False)])
def test_attention_forward_backward(hip_lib, B, L, H, dh, masked):
    from dfine_seg_amd import ops

    g = torch.Generator(device="cuda").manual_seed(L + dh)
    D = H * dh
    qk = (torch.randn(B, L, 2 * D, device="cuda", generator=g) * 1.5).bfloat16()
    v = torch.randn(B, L, D, device="cuda", generator=g).bfloat16()
    blocked = _dn_mask(L, Q=max(L - 60, L // 2)).cuda() if masked else None
    go = torch.randn(B, L, D, device="cuda", generator=g).bfloat16()

    qk_r, v_r = qk.float().requires_grad_(True), v.float().requires_grad_(True)
    ref = _ref(qk_r, v_r, blocked, H)
    ref.backward(go.float())

    qk_d, v_d = qk.clone().requires_grad_(True), v.clone().requires_grad_(True)
    assert ops.self_attention_supported(qk_d, v_d, H)
    out = ops.self_attention(qk_d, v_d, blocked, H)
    out.backward(go)
    torch.cuda.synchronize()
    _close(out, ref, "attention out")
    _close(v_d.grad, v_r.grad, "dv")
    _close(qk_d.grad[..., :D], qk_r.grad[..., :D], "dq")
    _close(qk_d.grad[..., D:], qk_r.grad[..., D:], "dk")


def test_fully_blocked_row_gives_zero_output_and_gradient(hip_lib):
    from dfine_seg_amd import ops

    g = torch.Generator(device="cuda").manual_seed(0)
    B, L, H, dh = 1, 70, 2, 32
    D = H * dh
    qk = torch.randn(B, L, 2 * D, device="cuda", generator=g).bfloat16().requires_grad_(True)
    v = torch.randn(B, L, D, device="cuda", generator=g).bfloat16().requires_grad_(True)
    blocked = torch.zeros(L, L, dtype=torch.bool, device="cuda")
    blocked[5, :] = True
    out = ops.self_attention(qk, v, blocked, H)
    out.sum().backward()
    assert torch.isfinite(out.float()).all() and float(out[0, 5].float().abs().max()) == 0.0
    assert torch.isfinite(qk.grad.float()).all() and float(qk.grad[0, 5, :D].float().abs().max()) == 0.0
    ref = _ref(qk.detach().float(), v.detach().float(), blocked, H)
    _close(out, ref, "blocked-row attention")


def test_reproducible_bit_for_bit(hip_lib):
    """No atomics anywhere: two runs give identical bits (forward and all three gradients)."""
    from dfine_seg_amd import ops

    g = torch.Generator(device="cuda").manual_seed(1)
    qk = torch.randn(2, 492, 512, device="cuda", generator=g).bfloat16()
    v = torch.randn(2, 492, 256, device="cuda", generator=g).bfloat16()
    go = torch.randn(2, 492, 256, device="cuda", generator=g).bfloat16()
    blocked = _dn_mask(492).cuda()
    res = []
    for _ in range(2):
        a, b = qk.clone().requires_grad_(True), v.clone().requires_grad_(True)
        o = ops.self_attention(a, b, blocked, 8)
        o.backward(go)
        res.append((o.detach().clone(), a.grad.clone(), b.grad.clone()))
    for x, y in zip(*res):
        assert torch.equal(x, y)


def test_packed_mask_equals_byte_mask(hip_lib):
    """The kernels read the mask bit-packed (one 64-bit word per lane and 64-row tile); with bits_q / bits_k = NULL they read
    `blocked` byte by byte. Same arithmetic: forward output, log-sum-exp and all three gradients must be BIT-equal, in the
    tiled and in the whole-sequence-resident builds. The packing itself is checked against an explicit bit loop."""
    from dfine_seg_amd import _lib, ops

    torch.manual_seed(5)
    B, L, H, dh = 2, 205, 4, 32  # ragged: 4 tiles, the last one 13 rows
    D = H * dh
    blocked = (torch.rand(L, L, device="cuda") < 0.3)
    blocked[7] = True  # a fully blocked row
    bits_q, bits_k = ops.attention_mask_bits(blocked)
    bq, bk, bl = bits_q.cpu(), bits_k.cpu(), blocked.cpu()
    for q in (0, 7, 63, 64, 204):
        for kk in (0, 1, 63, 64, 128, 204):
            assert bool((int(bq[q, kk // 64]) >> (kk % 64)) & 1) == bool(bl[q, kk])
            assert bool((int(bk[kk, q // 64]) >> (q % 64)) & 1) == bool(bl[q, kk])
    qk = torch.randn(B, L, 2 * D, device="cuda").bfloat16()
    v = torch.randn(B, L, D, device="cuda").bfloat16()
    go = torch.randn(B, L, D, device="cuda").bfloat16()
    mask = blocked.view(torch.uint8).contiguous()
    scale = 1.0 / dh ** 0.5

    def run(with_bits):
        o = torch.empty(B, L, D, dtype=torch.bfloat16, device="cuda")
        lse = torch.empty(B, H, L, dtype=torch.float32, device="cuda")
        dqk, dv, dsum = torch.empty_like(qk), torch.empty_like(v), torch.empty_like(lse)
        pq, pk = (_lib.ptr(bits_q), _lib.ptr(bits_k)) if with_bits else (None, None)
        rc = hip_lib.dfine_attention_fwd(_lib.ptr(qk), _lib.ptr(qk[..., D:]), _lib.ptr(v), 2 * D, 2 * D, D, _lib.ptr(mask), pq, _lib.ptr(o), D,
                                         _lib.ptr(lse), B, H, L, dh, scale, _lib.stream_ptr())
        assert rc == 0
        rc = hip_lib.dfine_attention_bwd(_lib.ptr(qk), _lib.ptr(qk[..., D:]), _lib.ptr(v), 2 * D, 2 * D, D, _lib.ptr(mask), pq, pk, _lib.ptr(o), D,
                                         _lib.ptr(lse), _lib.ptr(go), D, _lib.ptr(dqk), 2 * D, _lib.ptr(dqk[..., D:]), 2 * D, _lib.ptr(dv), D,
                                         _lib.ptr(dsum), B, H, L, dh, scale, _lib.stream_ptr())
        assert rc == 0
        torch.cuda.synchronize()
        return o, lse, dqk, dv

    a, b = run(False), run(True)
    for x, y, name in zip(a, b, ("o", "lse", "dqk", "dv")):
        assert torch.equal(x, y), name
    assert float(a[0][:, 7].float().abs().max()) == 0.0  # the blocked row
