"""Query-selection gather that shares the deformable-attention value-gradient sink (ops.select_rows):
gradients must equal the plain formulation (gather * keep, separate value gradients summed by autograd),
and a sink nobody finalizes must fail loudly instead of dropping the value gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(8, 8), (4, 4)]
NPTS = [4, 4]


def _inputs(dtype):
    g = torch.Generator(device="cuda").manual_seed(5)
    B, S, nH, c, Lq = 2, sum(h * w for h, w in SHAPES), 8, 32, 17
    memory = torch.randn(B, S, nH * c, device="cuda", generator=g).to(dtype)
    loc = torch.rand(B, Lq, nH, sum(NPTS), 2, device="cuda", generator=g)
    attn = torch.softmax(torch.randn(B, Lq, nH, sum(NPTS), device="cuda", generator=g), -1)
    idx = torch.stack([torch.randperm(S, device="cuda", generator=g)[:11] for _ in range(B)])
    keep = torch.rand(1, S, 1, device="cuda", generator=g) > 0.2
    return memory, loc, attn, idx, keep, (B, S, nH, c)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_select_rows_shares_the_value_gradient_sink(hip_lib, dtype):
    from dfine_seg_amd import ops

    memory, loc, attn, idx, keep, (B, S, nH, c) = _inputs(dtype)
    g = torch.Generator(device="cuda").manual_seed(6)
    w_rows = torch.randn(B, idx.shape[1], nH * c, device="cuda", generator=g).to(dtype)
    w_out = [torch.randn(B, loc.shape[1], nH * c, device="cuda", generator=g).to(dtype) for _ in range(2)]

    def run(shared):
        m = memory.clone().requires_grad_(True)
        value = m.view(B, S, nH, c)
        sink = ops.ValueGradSink() if shared else None
        if shared:
            rows = ops.select_rows(m, idx, keep, sink)
        else:
            rows = m.gather(1, idx.unsqueeze(-1).expand(-1, -1, nH * c)) * keep.view(-1)[idx].unsqueeze(-1).to(dtype)
        outs = [ops.deformable_attention_core(value, SHAPES, loc, attn, NPTS, sink, finalize=not shared) for _ in range(2)]
        loss = (rows * w_rows).float().sum() + sum((o * w).float().sum() for o, w in zip(outs, w_out))
        loss.backward()
        return rows.detach(), m.grad

    rows_a, grad_a = run(False)
    rows_b, grad_b = run(True)
    assert torch.equal(rows_a, rows_b)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=6e-2)
    torch.testing.assert_close(grad_b.float(), grad_a.float(), **tol)


def test_unfinalized_sink_fails_loudly(hip_lib):
    from dfine_seg_amd import ops

    memory, loc, attn, idx, keep, (B, S, nH, c) = _inputs(torch.float32)
    m = memory.clone().requires_grad_(True)
    sink = ops.ValueGradSink()
    out = ops.deformable_attention_core(m.view(B, S, nH, c), SHAPES, loc, attn, NPTS, sink, finalize=False)
    with pytest.raises(RuntimeError, match="never handed to autograd"):
        out.sum().backward()
    assert not sink.pending and sink.buffer is None


@pytest.mark.gpu
@pytest.mark.parametrize("shape,dtype", [((3, 8400, 80), torch.bfloat16), ((2, 525, 80), torch.float32), ((5, 37, 7), torch.bfloat16), ((1, 1, 300), torch.float32)])
def test_row_max_equals_torch(hip_lib, shape, dtype):
    """dfine_row_max (the best class logit per encoder token, whose top-k selects the decoder queries): exact."""
    from dfine_seg_amd import ops

    x = torch.randn(*shape, device="cuda").to(dtype)
    x[0, 0, -1] = 50.0
    with torch.no_grad():
        got = ops.row_max(x)
    assert got.dtype == torch.float32 and torch.equal(got, x.float().max(-1).values)
