"""ops.split_rows: the decoder's (denoising part, matching part) split of every head output (reference
src/d_fine/arch/dfine_decoder.py:1024-1050, torch.split on dn_num_split) as one autograd node whose backward is a single
concatenation. Pure PyTorch host logic: checked on the CPU against plain slicing."""
import torch


def _ops():
    from dfine_seg_amd import ops

    return ops


def test_split_rows_equals_slicing_forward_and_backward():
    ops = _ops()
    torch.manual_seed(0)
    t = torch.randn(3, 11, 5, requires_grad=True)
    r = t.detach().clone().requires_grad_()
    a, b = ops.split_rows(t, 4)
    assert torch.equal(a, r[:, :4]) and torch.equal(b, r[:, 4:])
    assert a.data_ptr() == t.data_ptr()  # views, no copies
    wa, wb = torch.randn(3, 4, 5), torch.randn(3, 7, 5)
    ((a * wa).sum() + (b * wb).sum() * 2).backward()
    ((r[:, :4] * wa).sum() + (r[:, 4:] * wb).sum() * 2).backward()
    assert torch.equal(t.grad, r.grad)


def test_split_rows_with_one_unused_part_and_with_an_empty_part():
    ops = _ops()
    t = torch.randn(2, 6, 3, requires_grad=True)
    a, b = ops.split_rows(t, 2)
    b.sum().backward()  # the denoising part receives no gradient: zeros there
    assert torch.equal(t.grad[:, :2], torch.zeros(2, 2, 3)) and torch.equal(t.grad[:, 2:], torch.ones(2, 4, 3))
    u = torch.randn(2, 6, 3, requires_grad=True)
    a, b = ops.split_rows(u, 0)  # no denoising queries in the batch
    assert a.shape == (2, 0, 3)
    (b * 3).sum().backward()
    assert torch.equal(u.grad, torch.full((2, 6, 3), 3.0))


def test_split_rows_without_grad_is_plain_slicing():
    ops = _ops()
    t = torch.randn(2, 5, 3)
    a, b = ops.split_rows(t, 2)
    assert a.grad_fn is None and b.grad_fn is None and torch.equal(torch.cat([a, b], 1), t)
