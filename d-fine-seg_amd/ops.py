"""Host-side operators over the C ABI: autograd Functions whose forward and backward are
HIP kernels from libdfine_hip.so. Names and argument meaning follow the reference's own
operator seams so the parity tests read like calls into the reference.
"""
import ctypes
import os
from typing import List, Sequence

import torch

from . import _lib


def msda_algorithmic_bytes(B, Lq, nH, P, c, elem):
    """(forward, backward) algorithmic bytes of one deformable-attention launch:
    fwd = 4 corner segments of c elements per sampling point + (x, y, weight) per point + the output;
    bwd = the same reads (value corners, loc, attn, grad_out) + one fp32 atomic segment per corner
    + grad_loc / grad_attn."""
    corners = B * Lq * nH * P * 4 * c
    side = B * Lq * nH * P * 12
    io = B * Lq * nH * c * elem
    return corners * elem + side + io, corners * elem + side + io + corners * 4 + side


class ZeroArena:
    """Pre-zeroed fp32 scratch handed out in slices: the fused ops need ~300 small zero-initialised
    workspaces per step (fp64 sum buffers, loss accumulators); one memset of the arena at the start of
    the step replaces one fill launch per workspace. ``ops.zero_arena`` is None unless a training loop
    installs one (``TrainStep`` owns its arena and installs it for the duration of its step); without it
    every workspace is its own ``torch.zeros``."""

    def __init__(self, numel, device):
        self.buf = torch.zeros(numel, dtype=torch.float32, device=device)
        self.used = 0

    def reset(self):
        """Zero the WHOLE arena and start handing out from the beginning. (Not just the extent earlier steps
        used: inside a captured graph the memset length is frozen at capture time, and a step captured later
        -- another ground-truth count, with masks -- may take more than any step before it; its tail would
        then keep accumulating across replays. 32 MB of memset is ~7 us.)"""
        self.buf.zero_()
        self.used = 0

    def take(self, n):
        n = (n + 3) // 4 * 4  # keep 16-byte alignment (fp64 views)
        if self.used + n > self.buf.numel():
            return None
        out = self.buf[self.used : self.used + n]
        self.used += n
        return out


zero_arena = None


def zeros_f32(n, device):
    if zero_arena is not None and zero_arena.buf.device == device:
        out = zero_arena.take(n)
        if out is not None:
            return out[:n]
    return torch.zeros(n, dtype=torch.float32, device=device)


class _MSDeformAttnCore(torch.autograd.Function):
    """Gather behind decoder cross-attention; see include/dfine_hip.h (dfine_msda_fwd/bwd)."""

    @staticmethod
    def forward(ctx, value, loc, attn, level_hw, num_points, sink=None, finalize=True):
        ctx.sink, ctx.finalize = sink, finalize
        _lib.require_gpu(value, loc, attn)
        lib = _lib.load()
        B, S, nH, c = value.shape
        Lq = loc.shape[1]
        P = sum(num_points)
        if loc.shape != (B, Lq, nH, P, 2) or attn.shape != (B, Lq, nH, P):
            raise ValueError(
                f"sampling_locations {tuple(loc.shape)} / attention_weights {tuple(attn.shape)} "
                f"do not match value {tuple(value.shape)} with {P} points per head"
            )
        value = value.contiguous()
        loc = loc.detach().float().contiguous()
        attn = attn.detach().float().contiguous()
        out = torch.empty(B, Lq, nH * c, dtype=value.dtype, device=value.device)
        hw = _lib.i32_array(v for pair in level_hw for v in pair)
        npts = _lib.i32_array(num_points)
        with _lib.timed("msda_fwd", msda_algorithmic_bytes(B, Lq, nH, P, c, value.element_size())[0]):
            rc = lib.dfine_msda_fwd(
                _lib.ptr(value), _lib.dtype_code(value), _lib.ptr(loc), _lib.ptr(attn), _lib.ptr(out),
                B, S, Lq, nH, c, len(num_points), hw, npts, _lib.stream_ptr(),
            )
        _lib.check(rc, "dfine_msda_fwd")
        ctx.save_for_backward(value, loc, attn)
        ctx.geom = (tuple(map(tuple, level_hw)), tuple(num_points))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        value, loc, attn = ctx.saved_tensors
        level_hw, num_points = ctx.geom
        lib = _lib.load()
        B, S, nH, c = value.shape
        Lq = loc.shape[1]
        grad_out = grad_out.to(value.dtype).contiguous()
        grad_loc = torch.empty_like(loc)
        grad_attn = torch.empty_like(attn)
        hw = _lib.i32_array(v for pair in level_hw for v in pair)
        npts = _lib.i32_array(num_points)
        sink = ctx.sink
        if MSDA_VALUE_GRAD == "atomic":
            # one kernel: grad_loc / grad_attn and the value gradient scattered with global float atomics into ONE fp32 buffer
            # shared by every call that samples this value tensor (measured faster than the LDS-slab variant below: the
            # chip's LDS float-atomic rate, ~0.18 T adds/s, is LOWER than its global one, ~0.31 T adds/s)
            if sink is not None and sink.buffer is not None:
                grad_value = sink.buffer
            else:
                # bf16 values (the training path): a bf16 map and PACKED bf16 atomics -- half the atomic bytes, the floor of
                # this kernel; the precision of stock autocast's grid_sample backward (DFINE_MSDA_GRAD=f32: float atomics)
                gdt = torch.bfloat16 if (value.dtype == torch.bfloat16 and MSDA_GRAD_BF16) else torch.float32
                grad_value = torch.zeros(B, S, nH, c, dtype=gdt, device=value.device)
                if sink is not None:
                    sink.buffer = grad_value
            with _lib.timed("msda_bwd", msda_algorithmic_bytes(B, Lq, nH, loc.shape[3], c, value.element_size())[1]):
                rc = lib.dfine_msda_bwd(
                    _lib.ptr(value), _lib.dtype_code(value), _lib.ptr(loc), _lib.ptr(attn), _lib.ptr(grad_out), _lib.ptr(grad_value),
                    _lib.dtype_code(grad_value), _lib.ptr(grad_loc), _lib.ptr(grad_attn), B, S, Lq, nH, c, len(num_points), hw, npts,
                    _lib.stream_ptr())
            _lib.check(rc, "dfine_msda_bwd")
            if sink is not None and not ctx.finalize:
                sink.expect_consumer()
                return None, grad_loc, grad_attn, None, None, None, None
            if sink is not None:
                sink.buffer = None
            return grad_value.to(value.dtype), grad_loc, grad_attn, None, None, None, None
        fwd_bytes = msda_algorithmic_bytes(B, Lq, nH, loc.shape[3], c, value.element_size())[0]
        with _lib.timed("msda_bwd_points", fwd_bytes + 2 * loc.numel() * 4):
            rc = lib.dfine_msda_bwd_points(
                _lib.ptr(value), _lib.dtype_code(value), _lib.ptr(loc), _lib.ptr(attn), _lib.ptr(grad_out),
                _lib.ptr(grad_loc), _lib.ptr(grad_attn), B, S, Lq, nH, c, len(num_points), hw, npts, _lib.stream_ptr())
        _lib.check(rc, "dfine_msda_bwd_points")
        # d(loss)/d(value): deferred to ONE atomic-free launch over every call that sampled this value tensor
        sink = ctx.sink if ctx.sink is not None else ValueGradSink()
        sink.add(value, loc, attn, grad_out, level_hw, num_points)
        if ctx.sink is not None and not ctx.finalize:
            sink.expect_consumer()
            return None, grad_loc, grad_attn, None, None, None, None
        return sink.flush(out_dtype=value.dtype).to(value.dtype), grad_loc, grad_attn, None, None, None, None


# How d(loss)/d(value) of the deformable attention is formed (DFINE_MSDA_VALUE):
#   "sorted" (default)  deferred: every call's backward computes grad_loc / grad_attn only and parks (loc, attn, grad_out); one
#                       dfine_msda_bwd_value_sorted launch per backward pass counting-sorts each (image, head)'s hits by token
#                       in LDS and gathers -- no float atomics, every element written once
#   "atomic"            global float atomics inside each call's backward kernel (193 M adds per layer: the chip's atomic rate)
#   "slab"              deferred, LDS slabs with LDS float atomics (dfine_msda_bwd_value; bitwise reproducible, slowest)
import os as _os

MSDA_VALUE_GRAD = _os.environ.get("DFINE_MSDA_VALUE", "sorted")
if MSDA_VALUE_GRAD not in ("atomic", "slab", "sorted"):
    raise ValueError("DFINE_MSDA_VALUE must be atomic, slab or sorted")
# "atomic" mode with bf16 values: float atomics into an fp32 map by default -- what the reference's AMP path does (grid_sample
# is on autocast's fp32 list, so its backward accumulates in fp32). DFINE_MSDA_GRAD=bf16 selects PACKED bf16 atomics into a
# bf16 map: half the atomic bytes (384 vs 630 us per layer at D-FINE-m batch 32) but the running sum is rounded to 8 bits at
# every add; where many sampling points pile onto a few memory tokens the sum swamps (tests/test_bf16_wellcond_gpu.py: backbone
# gradient cosine to the reference -0.5 with it, 0.61 with float atomics, in a fixture whose query boxes all coincide).
MSDA_GRAD_BF16 = _os.environ.get("DFINE_MSDA_GRAD", "f32") == "bf16"


class ValueGradSink:
    """d(loss)/d(value) when several deformable-attention calls sample the SAME value tensor (all decoder layers do,
    dfine_decoder.py:449): every call's backward computes its own grad_loc / grad_attn and parks (loc, attn, grad_out)
    here; ``flush()`` -- called by the call made FIRST in the forward pass (``finalize=True``: its backward runs last) or by
    ``select_rows`` -- turns all of them into the fp32 value gradient with ONE launch of dfine_msda_bwd_value (LDS slabs,
    no global atomics, every element written once)."""

    def __init__(self):
        self.pending = []   # "slab" mode: parked (loc, attn, grad_out) triples
        self.buffer = None  # "atomic" mode: the shared fp32 accumulator
        self.meta = None
        self._check_queued = False

    def add(self, value, loc, attn, grad_out, level_hw, num_points):
        meta = (tuple(value.shape), value.dtype, value.device, level_hw, num_points, loc.shape[1])
        if self.meta is not None and self.meta != meta:
            raise RuntimeError("ValueGradSink: calls with different value tensors / geometry share one sink")
        self.meta = meta
        self.pending.append((loc, attn, grad_out))

    def flush(self, out_dtype=torch.float32):
        """[B, S, nH, c] gradient of every call so far (None if there was none); empties the sink. fp32, or -- ``out_dtype``
        bfloat16, honoured by the sorted gather only -- the fp32 sums rounded once (anything else comes back fp32)."""
        if self.buffer is not None:
            total, self.buffer = self.buffer, None
            return total
        if not self.pending:
            return None
        (B, S, nH, c), dtype, device, level_hw, num_points, Lq = self.meta
        lib = _lib.load()
        hw = _lib.i32_array(v for pair in level_hw for v in pair)
        npts = _lib.i32_array(num_points)
        code = _lib.DT_BF16 if dtype == torch.bfloat16 else _lib.DT_F32
        P = sum(num_points)
        total = None
        if MSDA_VALUE_GRAD == "sorted" and len(self.pending) <= 8:
            n = len(self.pending)
            arr = lambda k: (ctypes.c_void_p * n)(*[t[k].data_ptr() for t in self.pending])
            odt = torch.bfloat16 if out_dtype == torch.bfloat16 else torch.float32
            total = torch.empty(B, S, nH, c, dtype=odt, device=device)
            cap = int(lib.dfine_msda_sorted_records(n, B, S, Lq, nH, P))
            entries = torch.empty(cap, 2, dtype=torch.int32, device=device)
            es = 2 if dtype == torch.bfloat16 else 4
            hits = n * B * Lq * nH * P * 4  # bound: every corner inside its map
            # bytes: points read twice (count, fill), one 8-byte record written and read per hit, a grad_out row per hit (from
            # L2), the gradient written once
            nbytes = 2 * n * B * Lq * nH * P * 12 + hits * (16 + c * es) + total.numel() * total.element_size()
            with _lib.timed("msda_bwd_value", nbytes):
                rc = lib.dfine_msda_bwd_value_sorted(code, arr(0), arr(1), arr(2), n, _lib.ptr(total), _lib.dtype_code(total), _lib.ptr(entries), cap,
                                                     B, S, Lq, nH, c, len(num_points), hw, npts, _lib.stream_ptr())
            _lib.check(rc, "dfine_msda_bwd_value_sorted")
            self.pending, self.meta = [], None
            return total
        for first in range(0, len(self.pending), 8):  # the kernel takes up to 8 calls per launch
            chunk = self.pending[first : first + 8]
            n = len(chunk)
            arr = lambda k: (ctypes.c_void_p * n)(*[t[k].data_ptr() for t in chunk])
            part = torch.empty(B, S, nH, c, dtype=torch.float32, device=device)
            nbytes = n * B * Lq * nH * (P * 12 + c * (2 if dtype == torch.bfloat16 else 4)) + part.numel() * 4
            with _lib.timed("msda_bwd_value", nbytes):
                rc = lib.dfine_msda_bwd_value(code, arr(0), arr(1), arr(2), n, _lib.ptr(part), B, S, Lq, nH, c, len(num_points), hw, npts,
                                              _lib.stream_ptr())
            if rc == -2:  # a level too wide for an LDS slab: the float-atomic kernel
                part.zero_()
                for loc, attn, go in chunk:
                    scratch_l, scratch_a = torch.empty_like(loc), torch.empty_like(attn)
                    dummy_value = torch.zeros(B, S, nH, c, dtype=dtype, device=device)
                    rc2 = lib.dfine_msda_bwd(_lib.ptr(dummy_value), code, _lib.ptr(loc), _lib.ptr(attn), _lib.ptr(go), _lib.ptr(part),
                                             _lib.DT_F32, _lib.ptr(scratch_l), _lib.ptr(scratch_a), B, S, Lq, nH, c, len(num_points), hw, npts,
                                             _lib.stream_ptr())
                    _lib.check(rc2, "dfine_msda_bwd")
            else:
                _lib.check(rc, "dfine_msda_bwd_value")
            total = part if total is None else total.add_(part)
        self.pending, self.meta = [], None
        return total

    def expect_consumer(self):
        """Called from a non-finalizing backward: somebody else (``select_rows``) must flush the sink before this backward
        pass ends. If nobody does the value gradient would be silently lost, so the end-of-backward callback raises."""
        if not self._check_queued:
            self._check_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._check)

    def _check(self):
        self._check_queued = False
        if self.pending or self.buffer is not None:
            self.pending, self.meta, self.buffer = [], None, None
            raise RuntimeError("deformable attention: d(loss)/d(value) was parked but never handed to autograd "
                               "(finalize=False on every call and no select_rows() on the backward path)")


class _SelectRows(torch.autograd.Function):
    """rows[b, j] = memory[b, idx[b, j]] * keep[idx[b, j]] -- the query-selection gather of the decoder
    memory (reference dfine_decoder.py:885-922 gathers AFTER running enc_output / the score head on all
    S tokens; only the selected rows ever receive a gradient, so the differentiable branch is run on the
    selected rows alone). The gradient is added to the fp32 ``sink`` the deformable-attention calls
    accumulate d(loss)/d(value) in -- memory IS that value tensor -- and this node, created before every
    decoder layer and therefore run after all of them in backward (the engine runs ready nodes in
    descending creation order), converts the total once."""

    @staticmethod
    def forward(ctx, memory, idx, keep, sink):
        B, S, D = memory.shape
        idx_e = idx.unsqueeze(-1).expand(-1, -1, D)
        scale = keep.view(-1)[idx].unsqueeze(-1).to(memory.dtype)  # [B, Q, 1]
        ctx.save_for_backward(idx_e, scale)
        ctx.sink, ctx.shape, ctx.dtype = sink, (B, S, D), memory.dtype
        return memory.gather(1, idx_e) * scale

    @staticmethod
    def backward(ctx, g):
        idx_e, scale = ctx.saved_tensors
        B, S, D = ctx.shape
        sink = ctx.sink
        # every deformable-attention call's value gradient, one launch -- in the memory's own dtype where the sink can form it
        # so (the sorted gather sums in fp32 registers and rounds once): autograd would add this node's rows to that gradient in
        # the memory's dtype anyway, and a bf16 map saves a conversion pass over [B, S, D]
        total = sink.flush(out_dtype=ctx.dtype) if sink is not None else None
        if total is None:
            total = torch.zeros(B, S, D, dtype=torch.float32, device=g.device)
        total.view(B, S, D).scatter_add_(1, idx_e, (g * scale).to(total.dtype))
        return total.view(B, S, D).to(ctx.dtype), None, None, None


def select_rows(memory, idx, keep, sink=None):
    """memory [B, S, D], idx [B, Q] int64 (distinct per image), keep [S] (0/1) -> [B, Q, D]. With a
    ``sink`` the deformable-attention calls on the same memory must all pass ``finalize=False``: this
    op hands the accumulated value gradient to autograd."""
    return _SelectRows.apply(memory, idx, keep, sink)


def deformable_attention_core(
    value: torch.Tensor,
    value_spatial_shapes: Sequence[Sequence[int]],
    sampling_locations: torch.Tensor,
    attention_weights: torch.Tensor,
    num_points_list: List[int],
    sink: "ValueGradSink" = None,
    finalize: bool = True,
) -> torch.Tensor:
    """MI355X replacement for ``deformable_attention_core_func_v2(..., method="default")``
    (reference src/d_fine/arch/utils.py:191-264).

    Same arguments as the reference except ``value``: the decoder memory viewed channel-last
    ``[B, sum(H_l*W_l), n_head, c]`` (float32 or bfloat16) instead of the reference's per-level
    tuple of ``(B, n_head, c, H_l*W_l)`` (use :func:`value_from_reference_layout` to convert).
    Returns ``[B, Lq, n_head*c]`` in ``value``'s dtype; fp32 accumulation inside.
    """
    if value.dtype not in (torch.float32, torch.bfloat16):
        value = value.float()
    return _MSDeformAttnCore.apply(
        value, sampling_locations, attention_weights,
        [tuple(int(v) for v in hw) for hw in value_spatial_shapes], [int(n) for n in num_points_list],
        sink, finalize,
    )


def value_from_reference_layout(value_levels) -> torch.Tensor:
    """Reference ``value_op`` output (tuple of ``(B, n_head, c, H_l*W_l)``,
    dfine_decoder.py:416-426) -> channel-last ``[B, S, n_head, c]``."""
    return torch.cat(list(value_levels), dim=-1).permute(0, 3, 1, 2).contiguous()


@torch.no_grad()
def hungarian_match(pred_logits, pred_boxes, tgt_labels, tgt_boxes, num_gt, alpha=0.25, gamma=2.0,
                    cost_class=2.0, cost_bbox=5.0, cost_giou=2.0):
    """MI355X replacement for ``HungarianMatcher.forward`` (reference src/d_fine/matcher.py:52-130)
    for ALL heads of a step at once, with nothing copied to the host.

    pred_logits [H, B, Q, C], pred_boxes [H, B, Q, 4] (cxcywh); tgt_labels int64 [B, G],
    tgt_boxes f32 [B, G, 4] padded; num_gt int32 [B] on the device.
    Returns (gt_of_query int32 [H, B, Q], query_of_gt int32 [H, B, G]); -1 = unmatched / padding.
    The reference's per-image (query_idx, gt_idx) lists are ``(query_of_gt[h, b, :n], arange(n))``.
    """
    _lib.require_gpu(pred_logits, pred_boxes, tgt_labels, tgt_boxes, num_gt)
    lib = _lib.load()
    H, B, Q, C = pred_logits.shape
    G = tgt_labels.shape[1]
    logits = pred_logits.float().contiguous()
    boxes = pred_boxes.float().contiguous()
    tgt_labels = tgt_labels.to(torch.int64).contiguous()
    tgt_boxes = tgt_boxes.float().contiguous()
    num_gt = num_gt.to(torch.int32).contiguous()
    dev = logits.device
    cost = torch.empty(H * B, G, Q, dtype=torch.float32, device=dev)
    query_of_gt = torch.empty(H, B, G, dtype=torch.int32, device=dev)
    gt_of_query = torch.empty(H, B, Q, dtype=torch.int32, device=dev)
    st = _lib.stream_ptr()
    rc = lib.dfine_match_cost(_lib.ptr(logits), _lib.ptr(boxes), _lib.ptr(tgt_labels), _lib.ptr(tgt_boxes),
                              _lib.ptr(num_gt), _lib.ptr(cost), H * B, B, Q, C, G, alpha, gamma,
                              cost_class, cost_bbox, cost_giou, st)
    _lib.check(rc, "dfine_match_cost")
    rc = lib.dfine_lsap(_lib.ptr(cost), _lib.ptr(num_gt), _lib.ptr(query_of_gt), _lib.ptr(gt_of_query),
                        H * B, B, G, Q, st)
    _lib.check(rc, "dfine_lsap")
    return gt_of_query, query_of_gt


@torch.no_grad()
def go_union(gt_of_query):
    """Dense form of ``DFINECriterion._get_go_indices`` (dfine_criterion.py:371-392):
    gt_of_query int32 [H, B, Q] -> go int32 [B, Q] (most frequent GT per query, -1 = none)."""
    _lib.require_gpu(gt_of_query)
    lib = _lib.load()
    H, B, Q = gt_of_query.shape
    gt_of_query = gt_of_query.contiguous()
    go = torch.empty(B, Q, dtype=torch.int32, device=gt_of_query.device)
    rc = lib.dfine_go_union(_lib.ptr(gt_of_query), _lib.ptr(go), H, B, Q, _lib.stream_ptr())
    _lib.check(rc, "dfine_go_union")
    return go


class _DepthwiseConv2d(torch.autograd.Function):
    """Channel-last depthwise conv; see include/dfine_hip.h (dfine_dwconv_*)."""

    @staticmethod
    def forward(ctx, x, weight, stride, stats=None):
        _lib.require_gpu(x, weight)
        lib = _lib.load()
        B, C, H, W = x.shape
        k = weight.shape[-1]
        xl = x.permute(0, 2, 3, 1)  # NHWC view of a channels_last tensor
        if not xl.is_contiguous():
            xl = xl.contiguous()
        w = weight.detach().float().reshape(C, k, k).contiguous()
        pad = (k - 1) // 2
        OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        y = torch.empty(B, OH, OW, C, dtype=xl.dtype, device=x.device)
        with _lib.timed("dwconv_fwd", (xl.numel() + y.numel()) * xl.element_size() + w.numel() * 4):
            rc = lib.dfine_dwconv_fwd(_lib.ptr(xl), _lib.ptr(w), _lib.ptr(y), _lib.dtype_code(xl),
                                      B, H, W, C, k, stride, _lib.ptr(stats), _lib.stream_ptr())
        _lib.check(rc, "dfine_dwconv_fwd")
        ctx.save_for_backward(xl, w)
        ctx.cfg = (stride, k, weight.dtype)
        ctx.weight = weight
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_y):
        xl, w = ctx.saved_tensors
        stride, k, w_dtype = ctx.cfg
        lib = _lib.load()
        B, H, W, C = xl.shape
        gy = grad_y.permute(0, 2, 3, 1).to(xl.dtype)
        if not gy.is_contiguous():
            gy = gy.contiguous()
        st = _lib.stream_ptr()
        code = _lib.dtype_code(xl)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(xl)
            with _lib.timed("dwconv_bwd_data", (gy.numel() + dx.numel()) * gy.element_size() + w.numel() * 4):
                rc = lib.dfine_dwconv_bwd_data(_lib.ptr(gy), _lib.ptr(w), _lib.ptr(dx), code, B, H, W, C, k, stride, st)
            _lib.check(rc, "dfine_dwconv_bwd_data")
            dx = dx.permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            wp = ctx.weight
            direct = (DIRECT_WGRAD_ACCUM and wp.grad is not None and wp.grad.dtype == torch.float32
                      and wp.grad.stride() in ((k * k, k * k, k, 1), (k * k, 1, k, 1)))  # memory order [C][k][k] either way
            dw = wp.grad if direct else torch.zeros(C, 1, k, k, dtype=torch.float32, device=xl.device)
            n_ws = lib.dfine_dwconv_bwd_weight_workspace(B, H, W, C, k, stride)
            if n_ws < 0:
                raise _lib.DfineHipError(f"depthwise conv weight gradient: unsupported shape C={C} k={k} stride={stride}")
            ws = torch.empty(n_ws, dtype=torch.float32, device=xl.device)
            with _lib.timed("dwconv_bwd_weight", (gy.numel() + xl.numel()) * gy.element_size() + w.numel() * 4):
                rc = lib.dfine_dwconv_bwd_weight(_lib.ptr(xl), _lib.ptr(gy), _lib.ptr(dw), _lib.ptr(ws), n_ws, code,
                                                 B, H, W, C, k, stride, st)
            _lib.check(rc, "dfine_dwconv_bwd_weight")
            dw = None if direct else dw.to(w_dtype)
        return dx, dw, None, None


def depthwise_conv2d(x: torch.Tensor, weight: torch.Tensor, stride: int = 1, stats=None) -> torch.Tensor:
    """MI355X replacement for ``F.conv2d(x, weight, None, stride, (k-1)//2, 1, groups=C)`` with
    ``weight`` [C, 1, k, k] (k in {3, 5}): x [B, C, H, W] in channels_last memory (float32 or
    bfloat16) -> y [B, C, OH, OW] channels_last, same dtype. The fp32 master weight is read
    directly (no autocast cast). ``stats`` (stride 1 only): as in ``dense_conv`` -- the kernel's epilogue leaves the
    following BatchNorm's statistics in it."""
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.float()
    return _DepthwiseConv2d.apply(x, weight, int(stride), stats)


def depthwise_stats_supported(stride):
    """Whether ``depthwise_conv2d`` can fill ``stats`` (the stride-1 strip kernel has the epilogue)."""
    return int(stride) == 1 and _os.environ.get("DFINE_DWCONV_NO_STRIP") is None


_ACT_CODES = {None: 0, "relu": 1, "silu": 2}


def _nhwc(t):
    v = t.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def _grad_rows(g, C):
    """NHWC gradient view [B, H, W, C] -> (tensor, pixel pitch) readable by the BN backward kernels in place: pixels at a
    constant pitch (dense, or a channel slice of a wider dense map) with 16-byte aligned channel vectors; else a dense copy."""
    es = g.element_size()
    vec = 16 // es
    B, H, W, _ = g.shape
    st = g.stride()
    if (st[3] == 1 and st[2] >= C and st[1] == W * st[2] and st[0] == H * st[1] and st[2] % vec == 0
            and g.data_ptr() % 16 == 0 and (C % vec == 0)):
        return g, st[2]
    return g.contiguous(), C


class ConcatBuffer:
    """One NHWC map [B, H, W, channels] that its producers fill by channel slice, in place of ``torch.cat`` (reference
    hgnetv2.py:258-266 HG_Block, hybrid_encoder.py:213-239 RepNCSPELAN4): a fused BatchNorm writes its output straight into its
    slice (``batch_norm_act(..., slot=(buffer, c0))``), other tensors are copied in (``concat_copy_in``), and ``concat_read``
    hands the whole map to the consumer. In backward the consumer's input gradient is parked here (``grad``) instead of
    being sliced through autograd: every producer reads its channel slice of it in place and adds it, on load, to the
    gradient it receives from its other consumer -- no cat, no slice copies, no gradient-accumulation add kernels."""

    def __init__(self, B, H, W, channels, dtype, device):
        self.buf = torch.empty(B, H, W, channels, dtype=dtype, device=device)
        self.grad = None
        self.used = 0

    def take(self, channels):
        c0 = self.used
        self.used += channels
        if self.used > self.buf.shape[-1]:
            raise _lib.DfineHipError("ConcatBuffer: more channels than reserved")
        return c0


class _ConcatCopyIn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, buffer, c0):
        C = x.shape[1]
        dst = buffer.buf[..., c0 : c0 + C]
        # through .data: slices other producers already returned are autograd-visible views of the same base, and an in-place
        # write that bumps ITS version would make autograd reject them (the slices are disjoint; ordering is ours to keep)
        buffer.buf.data[..., c0 : c0 + C].copy_(x.permute(0, 2, 3, 1))
        ctx.slot = (buffer, c0, C)
        ctx.set_materialize_grads(False)
        return dst.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        buffer, c0, C = ctx.slot
        out = buffer.grad[..., c0 : c0 + C].permute(0, 3, 1, 2) if buffer.grad is not None else None
        if g is not None:
            out = g if out is None else out + g
        return out, None, None


class _ConcatRead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, buffer, *parts):
        ctx.buffer = buffer
        return buffer.buf.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        gl = g.permute(0, 2, 3, 1)
        ctx.buffer.grad = gl if gl.is_contiguous() else gl.contiguous()
        return (None,) * (1 + len(ctx.needs_input_grad) - 1)


class _ConcatSlotSlice(torch.autograd.Function):
    """Channels [c0, c0 + C) of a slice a producer wrote into ``buffer`` (``y``, an autograd-visible view of it), for a second
    consumer that reads only PART of it (RepNCSPELAN4's chunk(2), hybrid_encoder.py:229-231). Backward adds that consumer's
    gradient into the parked concat gradient in place (one strided add over C channels) and hands nothing back through
    autograd: the producer finds the sum in its slice of ``buffer.grad`` -- no zero-padded gradient of the full slice, no
    accumulation add over it."""

    @staticmethod
    def forward(ctx, y, buffer, c0, C):
        ctx.slot = (buffer, c0, C)
        return buffer.buf[..., c0 : c0 + C].permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        buffer, c0, C = ctx.slot
        if g is not None:
            if buffer.grad is None:
                raise _lib.DfineHipError("concat_slot_slice: the buffer's reader has not run its backward yet")
            buffer.grad[..., c0 : c0 + C].add_(g.permute(0, 2, 3, 1))
        return None, None, None, None


def concat_slot_slice(y, buffer, c0, C):
    """``buffer`` channels [c0, c0 + C) as an NCHW-shaped view, tied to their producer's output ``y``; see _ConcatSlotSlice."""
    return _ConcatSlotSlice.apply(y, buffer, c0, C)


def concat_copy_in(x, buffer):
    """Copy ``x`` [B, C, H, W] into the next free channel slice of ``buffer``; returns the slice (an NCHW-shaped view)."""
    return _ConcatCopyIn.apply(x, buffer, buffer.take(x.shape[1]))


def concat_read(buffer, parts):
    """The filled buffer as one [B, channels, H, W] channels_last tensor. ``parts`` (every slice written into it) tie the
    producers into the autograd graph: their backward runs after this node's and finds the gradient in ``buffer.grad``."""
    if buffer.used != buffer.buf.shape[-1]:
        raise _lib.DfineHipError("concat_read: the buffer is not full")
    return _ConcatRead.apply(buffer, *parts)


def concat_buffer_supported(x, channel_counts):
    """bf16 mode on the GPU, every part a multiple of 8 channels (16-byte channel vectors at every slice offset)."""
    return (x.is_cuda and (torch.is_autocast_enabled() or x.dtype == torch.bfloat16)
            and all(c % 8 == 0 for c in channel_counts))


# off by default: measured on D-FINE-m 640 b32 the 50 served norms save 1.1 ms of reduction launches but their GEMMs' epilogues
# (x tile read + activation derivative per element, exposed after the K loop) cost 2.1 ms -- DESIGN.md section 3, "not kept"
BNRED_FUSION = os.environ.get("DFINE_BNRED", "0") == "1"


def _average_over_ranks(ws32, sync):
    """The fp64 sums of a norm workspace (handed around as their fp32 view) -> their average over the ranks of ``sync`` =
    (group, ranks). SUM + scale rather than AVG: gloo (the CPU-side rehearsal backend) has no AVG."""
    import torch.distributed as dist

    v = ws32.data.view(torch.float64)  # .data: the workspaces are views of one arena whose version counter saved tensors share
    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=sync[0])
    v.mul_(1.0 / sync[1])
bnred_counts = {"fused": 0, "discarded": 0}  # norm backwards served by a GEMM epilogue / epilogue sums that went unused
_last_bn_handle = None


class BnBackwardHandle:
    """Lets the ONE stride-1 dense conv that consumes a fused norm's output take that norm's backward reductions in its
    input-gradient GEMM's epilogue (dfine_igemm_bf16_bnred) -- the norm's backward then runs its dx pass only. Made by
    _BatchNormAct.forward, hung on the output tensor by ``batch_norm_act`` and picked up by ``dense_conv``.

    The sums are only valid when the gradient the norm's backward receives IS the tensor that launch wrote (autograd summed
    nothing onto it) and the concat-slice part (dy2) is the one the launch read: ``token`` records both, the norm's backward
    compares and otherwise zeroes the workspace and reduces by itself."""

    __slots__ = ("xl", "mean", "rstd", "g32", "b32", "ls32", "act", "slot", "consumers", "ws", "token")

    def __init__(self, xl, mean, rstd, g32, b32, ls32, act, slot):
        self.xl, self.mean, self.rstd, self.g32, self.b32, self.ls32, self.act, self.slot = xl, mean, rstd, g32, b32, ls32, act, slot
        self.consumers, self.ws, self.token = 0, None, None

    def slot_grad(self):
        C = self.xl.shape[-1]
        if self.slot is None or self.slot[0].grad is None:
            return None, 0
        return _grad_rows(self.slot[0].grad[..., self.slot[1] : self.slot[1] + C].to(self.xl.dtype), C)

    def begin(self, in_hw):
        """The BnRed for the consumer's input-gradient launch, or None when this norm cannot be served."""
        from . import gemm

        B, H, W, C = self.xl.shape
        if not BNRED_FUSION or self.consumers != 1 or self.token is not None or tuple(in_hw) != (H, W):
            return None
        gy2, _ = self.slot_grad()
        self.ws = zeros_f32(66 * C + 2, self.xl.device)
        self.token = ("open", (gy2.data_ptr(), gy2._version) if gy2 is not None else None)
        return gemm.BnRed(self.xl, gy2, self.mean, self.rstd, self.g32, self.b32, self.ls32, self.ws[: 64 * C], _ACT_CODES[self.act])

    def finish(self, dx_rows, done):
        if not done:  # shapes off the 16-byte paths: the plain launch ran and the workspace is untouched
            self.ws, self.token = None, ("plain",)
            return
        self.token = ("done", self.token[1], dx_rows.data_ptr(), dx_rows._version)

    def matches(self, gy, gy2):
        """True when ``gy`` / ``gy2`` (what the norm's backward is about to read) are what the fused launch summed over."""
        t = self.token
        return (t is not None and t[0] == "done" and t[1] == ((gy2.data_ptr(), gy2._version) if gy2 is not None else None)
                and t[2] == gy.data_ptr() and t[3] == gy._version)


class _BatchNormAct(torch.autograd.Function):
    """BN (batch statistics, or frozen/running statistics) + addend + activation + LAB; see
    include/dfine_hip.h (dfine_bn_act_*)."""

    @staticmethod
    def forward(ctx, x, addend, gamma, beta, running_mean, running_var, lab_scale, lab_bias, act, eps, momentum, training,
                ws=None, have_stats=False, slot=None, post=None, sync=None):
        _lib.require_gpu(x, gamma, beta)
        lib = _lib.load()
        B, C, H, W = x.shape
        xl = _nhwc(x)
        al = _nhwc(addend.to(x.dtype)) if addend is not None else None
        pl, ld_post = (None, 0) if post is None else _grad_rows(post.to(x.dtype).permute(0, 2, 3, 1), C)  # pixel rows, any pitch
        ctx.has_post = post is not None
        n_pix = B * H * W
        ldy = 0
        if slot is not None:  # write the output as a channel slice of a concat buffer (see ConcatBuffer)
            y = slot[0].buf[..., slot[1] : slot[1] + C]
            ldy = slot[0].buf.shape[-1]
            if y.dtype != xl.dtype or tuple(y.shape) != tuple(xl.shape):
                raise _lib.DfineHipError("batch_norm_act: the concat slot does not match the output")
        else:
            y = torch.empty_like(xl)
        ctx.slot = slot
        ctx.set_materialize_grads(False)
        # one zeroed workspace: fp64 sums [8 replicas][2C] (as 32C floats) | save_mean [C] | save_rstd [C]; with
        # ``have_stats`` the producing conv's epilogue (dfine_igemm_bf16) has already accumulated the sums into it
        if ws is None:
            ws, have_stats = zeros_f32(34 * C, x.device), False
        sums, save_mean, save_rstd = ws[: 32 * C], ws[32 * C : 33 * C], ws[33 * C :]
        g32, b32 = gamma.detach().float(), beta.detach().float()
        ls32 = lab_scale.detach().float() if lab_scale is not None else None
        lb32 = lab_bias.detach().float() if lab_bias is not None else None
        es = xl.element_size()
        # SyncBatchNorm (``sync`` = (process group, ranks)): the sums are averaged over the ranks between the two phases
        train_code = int(training) | ((sync[1] << 8) if (training and sync is not None) else 0)
        args = (_lib.ptr(xl), _lib.ptr(al), _lib.ptr(y), _lib.dtype_code(xl), _lib.ptr(sums), _lib.ptr(g32), _lib.ptr(b32),
                _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(save_mean), _lib.ptr(save_rstd), _lib.ptr(ls32),
                _lib.ptr(lb32), n_pix, C, _ACT_CODES[act], train_code, eps, momentum)
        rc = 0
        if training and not have_stats:
            with _lib.timed("bn_stats", xl.numel() * es, f"{n_pix}x{C}"):
                rc = lib.dfine_bn_act_fwd(*args, 1, ldy, None, 0, _lib.stream_ptr())
        if training and sync is not None:
            _average_over_ranks(sums, sync)
        with _lib.timed("bn_apply", (xl.numel() + y.numel() + (al.numel() if al is not None else 0) + (xl.numel() if pl is not None else 0)) * es,
                        f"{n_pix}x{C}"):
            rc = rc or lib.dfine_bn_act_fwd(*args, 2, ldy, _lib.ptr(pl), ld_post, _lib.stream_ptr())
        _lib.check(rc, "dfine_bn_act_fwd")
        if not training:  # frozen / eval statistics: the backward kernel reads them from the same two buffers
            save_mean = running_mean.detach().float()
            save_rstd = torch.rsqrt(running_var.detach().float() + eps)
        ctx.save_for_backward(xl, al, g32, b32, save_mean, save_rstd, ls32)
        ctx.cfg = (act, eps, train_code, gamma.dtype, lab_scale.dtype if lab_scale is not None else None)
        ctx.params = (gamma, beta, lab_scale, lab_bias)
        ctx.sync = sync if training else None
        global _last_bn_handle
        ctx.handle = None
        if (al is None and xl.dtype == torch.bfloat16 and C % 8 == 0 and BNRED_FUSION and ctx.sync is None
                and (training or gamma.requires_grad or beta.requires_grad)):
            ctx.handle = _last_bn_handle = BnBackwardHandle(xl, save_mean, save_rstd, g32, b32, ls32, act, slot)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_y):
        xl, al, g32, b32, save_mean, save_rstd, ls32 = ctx.saved_tensors
        act, eps, training, g_dtype, lab_dtype = ctx.cfg
        lib = _lib.load()
        n_pix, C = xl.numel() // xl.shape[-1], xl.shape[-1]
        # the output gradient: what autograd hands over (the next layer's input gradient), plus -- for an output that was
        # written into a concat buffer -- this layer's channel slice of the buffer's gradient, summed on load by the kernels
        gy2, ld2 = None, 0
        if ctx.slot is not None and ctx.slot[0].grad is not None:
            gy2, ld2 = _grad_rows(ctx.slot[0].grad[..., ctx.slot[1] : ctx.slot[1] + C].to(xl.dtype), C)
        if grad_y is None:
            if gy2 is None:
                return (None,) * 17
            gy, ld1, gy2, ld2 = gy2, ld2, None, 0
        else:
            gy, ld1 = _grad_rows(grad_y.to(xl.dtype).permute(0, 2, 3, 1), C)
        dpost = None
        if ctx.has_post:  # the residual's gradient is the output gradient itself (both parts of it)
            dpost = gy if gy2 is None else gy + gy2
            dpost = dpost.permute(0, 3, 1, 2)
        dx = torch.empty_like(xl)
        dadd = torch.empty_like(xl) if al is not None else None
        # one zeroed workspace: fp64 reductions [8 replicas][4C] (as 64C floats) | dgamma [C] | dbeta [C] | dlab [2]
        h, reduced = ctx.handle, False
        if h is not None and h.ws is not None and h.matches(gy, gy2):  # the consumer's input-gradient GEMM left the reductions
            ws, reduced = h.ws, True
            bnred_counts["fused"] += 1
        else:  # (sums of a gradient autograd has since added to are dropped with their workspace)
            ws = zeros_f32(66 * C + 2, xl.device)
            bnred_counts["discarded"] += int(h is not None and h.ws is not None)
        red, dgamma, dbeta, dlab = ws[: 64 * C], ws[64 * C : 65 * C], ws[65 * C : 66 * C], ws[66 * C :]
        gamma_p, beta_p, ls_p, lb_p = ctx.params
        plist = [gamma_p, beta_p] + ([ls_p, lb_p] if ls32 is not None else [])
        wants_param_grads = any(p.requires_grad for p in plist)  # FrozenBatchNorm2d: buffers, no LAB -> dx only
        direct = DIRECT_WGRAD_ACCUM and wants_param_grads and all(
            p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous() for p in plist)
        dls, dlb = dlab[0:1], dlab[1:2]
        if direct:  # accumulate straight into the fp32 .grad views (no AccumulateGrad add kernels)
            dgamma, dbeta = gamma_p.grad, beta_p.grad
            if ls32 is not None:
                dls, dlb = ls_p.grad, lb_p.grad
        es = xl.element_size()
        n_add = al.numel() if al is not None else 0
        n_dy = 2 if gy2 is not None else 1
        args = (_lib.ptr(xl), _lib.ptr(al), _lib.ptr(gy), ld1, _lib.ptr(gy2), ld2, _lib.dtype_code(xl), _lib.ptr(save_mean), _lib.ptr(save_rstd),
                _lib.ptr(g32), _lib.ptr(b32), _lib.ptr(ls32), _lib.ptr(red), _lib.ptr(dx), _lib.ptr(dadd), _lib.ptr(dgamma),
                _lib.ptr(dbeta), _lib.ptr(dls) if ls32 is not None else None, _lib.ptr(dlb) if ls32 is not None else None,
                n_pix, C, _ACT_CODES[act], training, eps)
        rc = 0
        if (training or wants_param_grads) and not reduced:  # the reductions feed the batch-statistics terms and the parameter gradients
            with _lib.timed("bn_bwd_reduce", ((2 + n_dy) * xl.numel() - xl.numel() + n_add) * es, f"{n_pix}x{C}"):
                rc = lib.dfine_bn_act_bwd(*args, 1, _lib.stream_ptr())
        if ctx.sync is not None:
            _average_over_ranks(red, ctx.sync)
        with _lib.timed("bn_bwd_dx", ((2 + n_dy) * xl.numel() + 2 * n_add) * es, f"{n_pix}x{C}"):
            rc = rc or lib.dfine_bn_act_bwd(*args, 2, _lib.stream_ptr())
        _lib.check(rc, "dfine_bn_act_bwd")
        if direct or not wants_param_grads:
            return (dx.permute(0, 3, 1, 2), dadd.permute(0, 3, 1, 2) if dadd is not None else None) + (None,) * 13 + (dpost, None)
        return (dx.permute(0, 3, 1, 2), dadd.permute(0, 3, 1, 2) if dadd is not None else None,
                dgamma.to(g_dtype), dbeta.to(g_dtype), None, None,
                dlab[0:1].to(lab_dtype) if ls32 is not None else None,
                dlab[1:2].to(lab_dtype) if ls32 is not None else None, None, None, None, None, None, None, None, dpost, None)


class _BatchNormPairAct(torch.autograd.Function):
    """y = act(bn_a(xa) + bn_b(xb)), both norms with batch statistics left by their convolutions' epilogues; see
    include/dfine_hip.h (dfine_bn_pair_act_*)."""

    @staticmethod
    def forward(ctx, xa, xb, norm_a, norm_b, act, ws_a, ws_b, slot):
        lib = _lib.load()
        B, C, H, W = xa.shape
        xal, xbl = _nhwc(xa), _nhwc(xb)
        n_pix, ldy = B * H * W, 0
        if slot is not None:
            y = slot[0].buf[..., slot[1] : slot[1] + C]
            ldy = slot[0].buf.shape[-1]
        else:
            y = torch.empty_like(xal)
        ctx.slot = slot
        ctx.set_materialize_grads(False)
        ga, ba, gb, bb = (t.detach().float() for t in (norm_a.weight, norm_a.bias, norm_b.weight, norm_b.bias))
        save_a, save_b = ws_a[32 * C : 34 * C], ws_b[32 * C : 34 * C]
        with _lib.timed("bn_apply", 3 * xal.numel() * xal.element_size(), f"pair {n_pix}x{C}"):
            rc = lib.dfine_bn_pair_act_fwd(_lib.ptr(xal), _lib.ptr(xbl), _lib.ptr(y), _lib.dtype_code(xal), _lib.ptr(ws_a), _lib.ptr(ws_b),
                                           _lib.ptr(ga), _lib.ptr(ba), _lib.ptr(gb), _lib.ptr(bb), _lib.ptr(norm_a.running_mean),
                                           _lib.ptr(norm_a.running_var), _lib.ptr(norm_b.running_mean), _lib.ptr(norm_b.running_var),
                                           _lib.ptr(save_a), _lib.ptr(save_b), n_pix, C, _ACT_CODES[act], 1, float(norm_a.eps),
                                           float(norm_a.momentum or 0.1), ldy, _lib.stream_ptr())
        _lib.check(rc, "dfine_bn_pair_act_fwd")
        ctx.save_for_backward(xal, xbl, ga, ba, gb, bb, save_a, save_b)
        ctx.cfg = (act, float(norm_a.eps))
        ctx.params = (norm_a.weight, norm_a.bias, norm_b.weight, norm_b.bias)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad_y):
        xal, xbl, ga, ba, gb, bb, save_a, save_b = ctx.saved_tensors
        act, eps = ctx.cfg
        lib = _lib.load()
        n_pix, C = xal.numel() // xal.shape[-1], xal.shape[-1]
        gy2, ld2 = None, 0
        if ctx.slot is not None and ctx.slot[0].grad is not None:
            gy2, ld2 = _grad_rows(ctx.slot[0].grad[..., ctx.slot[1] : ctx.slot[1] + C].to(xal.dtype), C)
        if grad_y is None:
            if gy2 is None:
                return (None,) * 8
            gy, ld1, gy2, ld2 = gy2, ld2, None, 0
        else:
            gy, ld1 = _grad_rows(grad_y.to(xal.dtype).permute(0, 2, 3, 1), C)
        dxa, dxb = torch.empty_like(xal), torch.empty_like(xbl)
        plist = ctx.params
        direct = DIRECT_WGRAD_ACCUM and all(p.requires_grad and p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous()
                                            for p in plist)
        ws = zeros_f32(48 * C + (0 if direct else 4 * C), xal.device)
        red = ws[: 48 * C]
        if direct:
            dga, dba, dgb, dbb = (p.grad for p in plist)
        else:
            dga, dba, dgb, dbb = (ws[48 * C + i * C : 48 * C + (i + 1) * C] for i in range(4))
        es = xal.element_size()
        n_dy = 2 if gy2 is not None else 1
        with _lib.timed("bn_bwd_dx", ((2 + n_dy) * 2 + 2) * xal.numel() * es, f"pair {n_pix}x{C}"):
            rc = lib.dfine_bn_pair_act_bwd(_lib.ptr(xal), _lib.ptr(xbl), _lib.ptr(gy), ld1, _lib.ptr(gy2), ld2, _lib.dtype_code(xal),
                                           _lib.ptr(save_a), _lib.ptr(save_b), _lib.ptr(ga), _lib.ptr(ba), _lib.ptr(gb), _lib.ptr(bb),
                                           _lib.ptr(red), _lib.ptr(dxa), _lib.ptr(dxb), _lib.ptr(dga), _lib.ptr(dba), _lib.ptr(dgb),
                                           _lib.ptr(dbb), n_pix, C, _ACT_CODES[act], 1, eps, _lib.stream_ptr())
        _lib.check(rc, "dfine_bn_pair_act_bwd")
        out = (dxa.permute(0, 3, 1, 2), dxb.permute(0, 3, 1, 2))
        if direct:
            return out + (None,) * 6
        # parameter gradients through autograd: hand them to the norms' parameters by hand (the modules are not tensors)
        with torch.no_grad():
            for p, gsrc in zip(plist, (dga, dba, dgb, dbb)):
                if p.requires_grad:
                    if p.grad is None:
                        p.grad = gsrc.to(p.dtype).clone()
                    else:
                        p.grad.add_(gsrc.to(p.grad.dtype))  # in place: .grad may be a view of a flat gradient buffer
        return out + (None,) * 6


def batch_norm_pair_act_supported(xa, xb, norm_a, norm_b, ws_a, ws_b):
    """Both branches are training-mode nn.BatchNorm2d on bf16 / fp32 CUDA maps of one shape whose batch statistics their
    convolutions' epilogues have already accumulated (``ws_*`` from ConvNorm._conv with have_stats)."""
    return (xa.is_cuda and xa.shape == xb.shape and xa.dtype == xb.dtype and xa.dtype in (torch.float32, torch.bfloat16)
            and isinstance(norm_a, torch.nn.BatchNorm2d) and isinstance(norm_b, torch.nn.BatchNorm2d) and norm_a.training and norm_b.training
            and ws_a is not None and ws_b is not None and norm_a.eps == norm_b.eps and (norm_a.momentum or 0.1) == (norm_b.momentum or 0.1)
            and os.environ.get("DFINE_BN_PAIR", "1") != "0")


def batch_norm_pair_act(xa, xb, norm_a, norm_b, act, ws_a, ws_b, slot=None):
    """``act(norm_a(xa) + norm_b(xb))`` for two training-mode BatchNorm2d whose statistics are already in ``ws_a`` / ``ws_b``
    (reference hybrid_encoder.py:115-121, VGGBlock.forward): one HIP launch forward, two backward."""
    return _BatchNormPairAct.apply(xa, xb, norm_a, norm_b, act, ws_a, ws_b, slot)


def bn_workspace(channels, device):
    """Zeroed workspace of one fused BatchNorm call; its first 32*C floats are the fp64 statistics replicas a producing
    conv may fill (``dense_conv(..., stats=ws[:32 * C])`` then ``batch_norm_act(..., ws=ws, have_stats=True)``)."""
    return zeros_f32(34 * channels, device)


def batch_norm_act(x, gamma, beta, running_mean, running_var, act=None, lab_scale=None, lab_bias=None,
                   addend=None, training=True, momentum=0.1, eps=1e-5, ws=None, have_stats=False, slot=None, post=None, sync=None):
    """Fused ``lab_scale * act(batch_norm(x) + addend) + lab_bias (+ post)`` on a channels_last map
    (float32 or bfloat16). ``lab_scale`` / ``lab_bias`` are float tensors [1] (both or neither);
    ``act`` is None, "relu" or "silu". In training mode the running statistics are updated in place exactly like
    nn.BatchNorm2d (momentum, unbiased variance). ``sync`` = (process group or None for the default one, ranks): the batch
    statistics are those of all ranks' batches (nn.SyncBatchNorm; every rank must hold the same number of pixels): one small
    all-reduce of the fp64 sums per direction, see include/dfine_hip.h (dfine_bn_act_fwd, training | R << 8)."""
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.float()
    global _last_bn_handle
    _last_bn_handle = None
    y = _BatchNormAct.apply(x, addend, gamma, beta, running_mean, running_var, lab_scale, lab_bias, act, float(eps),
                            float(momentum), bool(training), ws, bool(have_stats), slot, post, sync)
    if _last_bn_handle is not None and y.requires_grad:
        y._bn_handle = _last_bn_handle  # see BnBackwardHandle
    _last_bn_handle = None
    return y


def batch_norm_act_supported(x, channels):
    """Any channel count (narrow vectors for odd counts, e.g. D-FINE-n's 21-channel CSP branch)."""
    return x.is_cuda and x.dtype in (torch.float32, torch.bfloat16)


class _SplitRows(torch.autograd.Function):
    """(t[:, :n], t[:, n:]) of a [B, L, ...] tensor as ONE autograd node: the decoder hands the denoising and the matching part
    of every head output to the criterion (reference dfine_decoder.py:1024-1050, torch.split on dn_num_split), and two
    separate slices cost autograd a zero-filled full tensor + a copy each and an add of the two (5 launches per tensor, ~16
    tensors per step); this node's backward is one concatenation."""

    @staticmethod
    def forward(ctx, t, n):
        ctx.n, ctx.shape, ctx.meta = n, t.shape, (t.dtype, t.device)
        ctx.set_materialize_grads(False)
        return t[:, :n], t[:, n:]

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None and gb is None:
            return None, None
        B, L = ctx.shape[0], ctx.shape[1]
        rest = tuple(ctx.shape[2:])
        if ga is None:
            ga = torch.zeros((B, ctx.n) + rest, dtype=gb.dtype, device=gb.device)
        if gb is None:
            gb = torch.zeros((B, L - ctx.n) + rest, dtype=ga.dtype, device=ga.device)
        return torch.cat([ga, gb.to(ga.dtype)], 1), None


def split_rows(t, n):
    """``(t[:, :n], t[:, n:])`` (views) whose gradients come back as one concatenation (see _SplitRows)."""
    if not (torch.is_grad_enabled() and t.requires_grad):
        return t[:, :n], t[:, n:]
    return _SplitRows.apply(t, int(n))


criterion_batch = None  # an open CriterionBatch, or None: every criterion_head call launches by itself
CRITERION_BATCH = os.environ.get("DFINE_CRITERION_BATCH", "1") != "0"  # (A/B measurements)


class CriterionBatch:
    """Collects the heads of one criterion call so that ONE dfine_criterion_heads launch computes them all (their kernels are
    independent and each leaves most of the chip idle). ``with CriterionBatch():`` around the ``criterion_head`` calls; the
    terms they return are filled when the block ends -- nothing may read them before."""

    def __init__(self):
        self.heads, self.keep, self.dtype, self.nbytes = [], [], None, 0

    def add(self, args, dtype, tensors, nbytes):
        self.heads.append(args)
        self.keep.append(tensors)
        self.dtype = dtype
        self.nbytes += nbytes

    def __enter__(self):
        global criterion_batch
        if criterion_batch is not None:
            raise _lib.DfineHipError("CriterionBatch: already open")
        criterion_batch = self
        return self

    def __exit__(self, exc_type, exc, tb):
        global criterion_batch
        criterion_batch = None
        if exc_type is None and self.heads:
            self.launch()
        self.heads, self.keep = [], []
        return False

    def launch(self):
        lib = _lib.load()
        n = len(self.heads)
        arr = (_lib.CriterionHeadArgs * n)(*self.heads)
        with _lib.timed("criterion_head", self.nbytes, f"{n} heads"):
            rc = lib.dfine_criterion_heads(ctypes.cast(arr, ctypes.c_void_p), n, _lib.DT_BF16 if self.dtype == torch.bfloat16 else _lib.DT_F32,
                                           _lib.stream_ptr())
        _lib.check(rc, "dfine_criterion_heads")


class _CriterionHead(torch.autograd.Function):
    """Weighted loss terms of one head + their gradients; see include/dfine_hip.h (dfine_criterion_head)."""

    @staticmethod
    def forward(ctx, logits, boxes, corners, ref_points, teacher_corners, teacher_logits, own, go, tgt_labels,
                tgt_boxes, Wn, scalars, alpha, gamma, reg_scale, temperature):
        _lib.require_gpu(logits, boxes, own, go)
        lib = _lib.load()
        B, Q, C = logits.shape
        G = tgt_labels.shape[1]
        dev = logits.device
        if logits.dtype not in (torch.float32, torch.bfloat16):
            logits = logits.float()
        strides = [0] * 6  # elements between images of each input, 0 = dense (dfine_criterion_head batch_strides)

        def rows(t, slot, dtype=None):
            """[B, Q, W] input as the kernel reads it: the [:, a:b] part of a wider decoder output stays in place (dense rows
            per image, a batch stride), anything else becomes a dense copy."""
            if dtype is not None and t.dtype != dtype:
                t = t.to(dtype)
            Wd = t.shape[-1]
            if t.dim() == 3 and t.stride(2) == 1 and t.stride(1) == Wd and t.stride(0) >= t.shape[1] * Wd and (
                    t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 or slot not in (1, 3)):
                strides[slot] = t.stride(0)
                return t
            return t.contiguous()

        logits, boxes = rows(logits, 0), rows(boxes, 1, torch.float32)
        has_local = corners is not None
        has_ddf = has_local and teacher_corners is not None
        n_bins = corners.shape[-1] // 4 if has_local else 0
        if has_local:
            corners, ref_points = rows(corners, 2, torch.float32), rows(ref_points, 3, torch.float32)
        if has_ddf:
            teacher_corners = rows(teacher_corners, 4, torch.float32)
            teacher_logits = rows(teacher_logits, 5, logits.dtype)
        terms = zeros_f32(64 * 8, dev)  # 64 replicas of the five sums (see dfine_criterion_head)
        d_logits = torch.empty(logits.shape, dtype=logits.dtype, device=dev)  # outputs are dense whatever the inputs' pitch
        d_l1 = torch.empty(B, Q, 4, dtype=torch.float32, device=dev)
        d_giou = torch.empty_like(d_l1)
        d_fgl = torch.empty(corners.shape, dtype=torch.float32, device=dev) if has_local else None
        d_ddf = torch.empty(corners.shape, dtype=torch.float32, device=dev) if has_ddf else None
        batch = criterion_batch
        if batch is not None and (not batch.heads or batch.dtype == logits.dtype):
            # a CriterionBatch is open (DFINECriterion.forward): park the head; ONE launch computes every head when it closes,
            # before anybody reads the terms returned here
            out = torch.empty(5, dtype=torch.float32, device=dev)
            a = _lib.CriterionHeadArgs(
                logits=logits.data_ptr(), boxes=boxes.data_ptr(), corners=corners.data_ptr() if has_local else None,
                ref_points=ref_points.data_ptr() if has_local else None, teacher_corners=teacher_corners.data_ptr() if has_ddf else None,
                teacher_logits=teacher_logits.data_ptr() if has_ddf else None, own_assign=own.data_ptr(), go_assign=go.data_ptr(),
                tgt_labels=tgt_labels.data_ptr(), tgt_boxes=tgt_boxes.data_ptr(), Wn=Wn.data_ptr() if Wn is not None else None,
                scalars=scalars.data_ptr(), terms=terms.data_ptr(), d_logits=d_logits.data_ptr(), d_boxes_l1=d_l1.data_ptr(),
                d_boxes_giou=d_giou.data_ptr(), d_corners_fgl=d_fgl.data_ptr() if has_local else None,
                d_corners_ddf=d_ddf.data_ptr() if has_ddf else None, out=out.data_ptr(), B=B, Q=Q, C=C, G=G, n_bins=n_bins, has_strides=1,
                alpha=alpha, gamma=gamma, reg_scale=reg_scale, temperature=temperature, batch_strides=(ctypes.c_longlong * 6)(*strides))
            batch.add(a, logits.dtype, (logits, boxes, corners, ref_points, teacher_corners, teacher_logits, own, go, tgt_labels, tgt_boxes, Wn,
                                        scalars, terms, d_logits, d_l1, d_giou, d_fgl, d_ddf, out),
                      logits.numel() * logits.element_size() * 2 + (corners.numel() * 16 if has_local else 0))
            ctx.save_for_backward(d_logits, d_l1, d_giou, d_fgl, d_ddf)
            ctx.mark_non_differentiable()
            return tuple(out.unbind(0))
        with _lib.timed("criterion_head", logits.numel() * logits.element_size() * 2 + (corners.numel() * 16 if has_local else 0)):
            rc = lib.dfine_criterion_head(
                _lib.ptr(logits), _lib.dtype_code(logits), _lib.ptr(boxes), _lib.ptr(corners), _lib.ptr(ref_points),
                _lib.ptr(teacher_corners) if has_ddf else None, _lib.ptr(teacher_logits) if has_ddf else None,
                _lib.ptr(own), _lib.ptr(go), _lib.ptr(tgt_labels), _lib.ptr(tgt_boxes), _lib.ptr(Wn), _lib.ptr(scalars),
                _lib.ptr(terms), _lib.ptr(d_logits), _lib.ptr(d_l1), _lib.ptr(d_giou), _lib.ptr(d_fgl), _lib.ptr(d_ddf),
                B, Q, C, G, n_bins, alpha, gamma, reg_scale, temperature, (ctypes.c_longlong * 6)(*strides), _lib.stream_ptr())
        _lib.check(rc, "dfine_criterion_head")
        ctx.save_for_backward(d_logits, d_l1, d_giou, d_fgl, d_ddf)
        ctx.mark_non_differentiable()
        return tuple(terms.view(64, 8)[:, :5].sum(0).unbind(0))

    @staticmethod
    def backward(ctx, g_vfl, g_l1, g_giou, g_fgl, g_ddf):
        d_logits, d_l1, d_giou, d_fgl, d_ddf = ctx.saved_tensors
        lib = _lib.load()
        grad_logits, grad_boxes = torch.empty_like(d_logits), torch.empty_like(d_l1)
        grad_corners = torch.empty_like(d_fgl) if d_fgl is not None else None
        gs = [g.float().reshape(1) if g is not None else None for g in (g_vfl, g_l1, g_giou, g_fgl, g_ddf)]
        rc = lib.dfine_criterion_head_scale(
            _lib.ptr(d_logits), _lib.dtype_code(d_logits), _lib.ptr(d_l1), _lib.ptr(d_giou), _lib.ptr(d_fgl), _lib.ptr(d_ddf),
            *[_lib.ptr(g) for g in gs], _lib.ptr(grad_logits), _lib.ptr(grad_boxes), _lib.ptr(grad_corners),
            d_logits.numel(), d_l1.numel(), d_fgl.numel() if d_fgl is not None else 0, _lib.stream_ptr())
        _lib.check(rc, "dfine_criterion_head_scale")
        return (grad_logits, grad_boxes, grad_corners) + (None,) * 13


def criterion_head(logits, boxes, corners, ref_points, teacher_corners, teacher_logits, own_assign, go_assign,
                   tgt_labels, tgt_boxes, Wn, scalars, alpha=0.75, gamma=2.0, reg_scale=4.0, temperature=5.0):
    """Weighted (loss_vfl, loss_bbox, loss_giou, loss_fgl, loss_ddf) of one prediction head as five
    0-dim tensors, differentiable w.r.t. logits, boxes and corners. ``corners`` None: no FGL/DDF
    (their terms are 0); ``teacher_*`` None: no DDF. ``scalars``: device float [9], see dfine_hip.h."""
    return _CriterionHead.apply(logits, boxes, corners, ref_points, teacher_corners, teacher_logits,
                                own_assign.contiguous(), go_assign.contiguous(), tgt_labels.contiguous(),
                                tgt_boxes.contiguous(), Wn, scalars, float(alpha), float(gamma), float(reg_scale),
                                float(temperature))


class _StemPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slot):
        _lib.require_gpu(x)
        lib = _lib.load()
        xl = _nhwc(x)
        B, H, W, C = xl.shape
        ldy = 0
        if slot is not None:  # the pooled map is a channel slice of a concat buffer
            y = slot[0].buf[..., slot[1] : slot[1] + C]
            ldy = slot[0].buf.shape[-1]
            if y.dtype != xl.dtype or tuple(y.shape) != tuple(xl.shape):
                raise _lib.DfineHipError("stem_pool: the concat slot does not match the output")
        else:
            y = torch.empty_like(xl)
        with _lib.timed("stem_pool_fwd", 2 * xl.numel() * xl.element_size()):
            rc = lib.dfine_stem_pool_fwd(_lib.ptr(xl), _lib.dtype_code(xl), _lib.ptr(y), B, H, W, C, ldy, _lib.stream_ptr())
        _lib.check(rc, "dfine_stem_pool_fwd")
        ctx.save_for_backward(xl)
        ctx.slot = slot
        ctx.set_materialize_grads(False)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        (xl,) = ctx.saved_tensors
        lib = _lib.load()
        B, H, W, C = xl.shape
        if ctx.slot is not None and ctx.slot[0].grad is not None:  # this map's slice of the concat's gradient, read in place
            g2 = ctx.slot[0].grad[..., ctx.slot[1] : ctx.slot[1] + C].to(xl.dtype)
            gy = g2 if gy is None else g2 + gy.permute(0, 2, 3, 1)
        elif gy is None:
            return None, None
        else:
            gy = gy.to(xl.dtype).permute(0, 2, 3, 1)
        gyl, ld = _grad_rows(gy, C)
        gx = torch.empty_like(xl)
        with _lib.timed("stem_pool_bwd", 3 * xl.numel() * xl.element_size()):
            rc = lib.dfine_stem_pool_bwd(_lib.ptr(xl), _lib.ptr(gyl), ld, _lib.dtype_code(xl), _lib.ptr(gx), B, H, W, C, _lib.stream_ptr())
        _lib.check(rc, "dfine_stem_pool_bwd")
        return gx.permute(0, 3, 1, 2), None


def stem_pool(x, into=None):
    """``F.max_pool2d(F.pad(x, (0, 1, 0, 1)), 2, stride=1, ceil_mode=True)`` on a channels_last map
    [B, C, H, W] -> [B, C, H, W] (HGNetv2 stem, hgnetv2.py:213-224), HIP, no padded copy. ``into`` (a ConcatBuffer): the
    result becomes the buffer's next channel slice."""
    return _StemPool.apply(x, (into, into.take(x.shape[1])) if into is not None else None)


class _MaskLogits(torch.autograd.Function):
    """pred[b, m, hw] = sum_c rows[b, m, c] * feat[b, c, hw] for a CHANNELS-LAST feature map, with the feature-map gradient
    produced channels-last too (g^T @ rows -> [B, HW, C]): einsum's own backward leaves it as a dense [B, C, HW] map, which
    the norm kernel behind it then has to transpose (a 0.67 ms copy of a 420 MB map at 640^2, batch 32)."""

    @staticmethod
    def forward(ctx, rows, feat):
        B, C, H, W = feat.shape
        fl = _nhwc(feat).view(B, H * W, C)
        ctx.save_for_backward(rows, fl)
        ctx.shape = (B, C, H, W)
        return torch.bmm(rows, fl.transpose(1, 2)).view(B, rows.shape[1], H, W)

    @staticmethod
    def backward(ctx, g):
        rows, fl = ctx.saved_tensors
        B, C, H, W = ctx.shape
        g = g.reshape(B, rows.shape[1], H * W)
        d_rows = torch.bmm(g, fl) if ctx.needs_input_grad[0] else None
        d_feat = torch.bmm(g.transpose(1, 2), rows).view(B, H, W, C).permute(0, 3, 1, 2) if ctx.needs_input_grad[1] else None
        return d_rows, d_feat


def mask_logits_rows(rows, feat):
    """``einsum("bmc,bchw->bmhw", rows, feat)`` (the matched queries' mask logits, dfine_criterion.py:272-357 /
    dfine_decoder.py:937-940); for a channels_last CUDA feature map the gradient comes back channels_last (see _MaskLogits)."""
    rows = rows.to(feat.dtype)
    if feat.is_cuda and feat.dim() == 4 and feat.permute(0, 2, 3, 1).is_contiguous():
        return _MaskLogits.apply(rows, feat)
    return torch.einsum("bmc,bchw->bmhw", rows, feat)


def _grouped_wgrad_now(problems, keep, nbytes, flops):
    """Launch a list of _lib.WgradProblem as ONE grouped launch right now (include/dfine_hip.h: dfine_wgrad_group_*),
    outside the step's deferred queue: the results are needed by this backward pass. Under hipGraph capture the launch table
    takes persistent memory of the installed queue (see WgradQueue); without one the problems are launched singly."""
    lib = _lib.load()
    n = len(problems)
    arr = (_lib.WgradProblem * n)(*problems)
    capturing = torch.cuda.is_current_stream_capturing()
    need = lib.dfine_wgrad_group_plan(arr, n, None, 0) if all(lib.dfine_wgrad_group_eligible(ctypes.byref(q)) for q in problems) else 0
    dev = None
    if need > 0:
        if capturing:
            dev = wgrad_queue._persistent(int(need)) if wgrad_queue is not None else None
        else:
            dev = torch.empty(int(need), dtype=torch.uint8, device=keep[0].device)
    if dev is None:
        for q in problems:
            _lib.check(lib.dfine_linear_wgrad(q.dY, q.X, q.dW, q.db, q.M, q.N, q.K, q.ldw, q.ldy, q.ldx, _lib.stream_ptr()), "dfine_linear_wgrad")
        return
    host = torch.empty(int(need), dtype=torch.uint8, pin_memory=not capturing)
    used = lib.dfine_wgrad_group_plan(arr, n, ctypes.c_void_p(host.data_ptr()), int(need))
    if used != need:
        _lib.check(int(used) if used < 0 else -1, "dfine_wgrad_group_plan")
    if capturing:
        wgrad_queue.pending_uploads.append((dev, host))
        wgrad_queue.max_need = max(wgrad_queue.max_need, int(need))
    else:
        dev.copy_(host, non_blocking=True)
    with _lib.timed("wgrad", nbytes, f"grouped x{n} (mask rows)", flops=flops):
        rc = lib.dfine_wgrad_group_launch(ctypes.c_void_p(host.data_ptr()), _lib.ptr(dev), _lib.stream_ptr())
    _lib.check(rc, "dfine_wgrad_group_launch")
    if not capturing:
        host_keepalive.append((host, torch.cuda.current_stream().record_event()))
        while host_keepalive and host_keepalive[0][1].query():
            host_keepalive.pop(0)


host_keepalive = []  # pinned launch tables whose async upload may still be in flight


class _MaskRowsLoss(torch.autograd.Function):
    """stats[b, n] = (sum focal terms, sum p t, sum p) of the mask logits rows[b, n] . feat[b, :, px] against ground truth
    n % G -- the GEMM and the loss statistics in ONE kernel each way (include/dfine_hip.h: dfine_mask_rows_*): the
    [B, N, HW] logits (680 MB at D-FINE-m 640^2 batch 32) are never written. Backward: the logits are recomputed tile by
    tile and turned into d(loss)/d(logit) in the accumulators (bf16 [B*HW, N]), which feeds the two gradient GEMMs."""

    @staticmethod
    def forward(ctx, rows, feat, tgt, alpha):
        lib = _lib.load()
        B, C, H, W = feat.shape
        N, G, HW = rows.shape[1], tgt.shape[1], H * W
        fl = _nhwc(feat).reshape(B * HW, C)
        r = rows.to(torch.bfloat16).contiguous()
        t, a = tgt.float().contiguous(), alpha.float().contiguous()
        stats = zeros_f32(B * N * 3, fl.device).view(B, N, 3)
        with _lib.timed("mask_rows_stats", (fl.numel() + r.numel()) * 2 + t.numel() * 4, flops=2.0 * B * HW * N * C):
            rc = lib.dfine_mask_rows_stats(_lib.ptr(fl), _lib.ptr(r), _lib.ptr(t), _lib.ptr(a), _lib.ptr(stats), B, HW, N, G, C, _lib.stream_ptr())
        _lib.check(rc, "dfine_mask_rows_stats")
        ctx.save_for_backward(r, fl, t, a)
        ctx.shape = (B, C, H, W, rows.dtype)
        return stats

    @staticmethod
    def backward(ctx, g_stats):
        r, fl, t, a = ctx.saved_tensors
        B, C, H, W, rows_dtype = ctx.shape
        lib = _lib.load()
        N, G, HW = r.shape[1], t.shape[1], H * W
        Npad = (N + 7) // 8 * 8
        g = g_stats.float().contiguous()
        dz = torch.empty(B * HW, Npad, dtype=torch.bfloat16, device=fl.device)
        if Npad != N:
            dz[:, N:].zero_()
        with _lib.timed("mask_rows_dz", (fl.numel() + r.numel() + dz.numel()) * 2 + t.numel() * 4, flops=2.0 * B * HW * N * C):
            rc = lib.dfine_mask_rows_dz(_lib.ptr(fl), _lib.ptr(r), _lib.ptr(t), _lib.ptr(a), _lib.ptr(g), _lib.ptr(dz), Npad, B, HW, N, G, C,
                                        _lib.stream_ptr())
        _lib.check(rc, "dfine_mask_rows_dz")
        d_rows = d_feat = None
        if ctx.needs_input_grad[1]:
            rows_t = torch.zeros(B, C, Npad, dtype=torch.bfloat16, device=fl.device)
            rows_t[:, :, :N] = r.transpose(1, 2)
            d_feat = torch.empty(B * HW, C, dtype=torch.bfloat16, device=fl.device)
            with _lib.timed("gemm_dgrad_rows", (dz.numel() + d_feat.numel() + rows_t.numel()) * 2, "mask d_feat", flops=2.0 * B * HW * Npad * C):
                rc = lib.dfine_mask_rows_dfeat(_lib.ptr(dz), Npad, _lib.ptr(rows_t), Npad, _lib.ptr(d_feat), B, HW, Npad, C, _lib.stream_ptr())
            _lib.check(rc, "dfine_mask_rows_dfeat")
            d_feat = d_feat.view(B, H, W, C).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[0]:
            # d_rows[b] [N, C] = dz[b]^T [N, HW] @ feat[b] [HW, C]: weight-gradient shaped (reduction over the pixel rows)
            dr = zeros_f32(B * N * C, fl.device).view(B, N, C) if B * N * C <= (1 << 20) else torch.zeros(B, N, C, dtype=torch.float32, device=fl.device)
            probs = [_lib.WgradProblem(dY=dz.data_ptr() + b * HW * Npad * 2, X=fl.data_ptr() + b * HW * C * 2, dW=dr.data_ptr() + b * N * C * 4,
                                       db=None, conv=0, M=HW, N=N, K=C, ldw=C, ldy=Npad, ldx=C) for b in range(B)]
            _grouped_wgrad_now(probs, (dz, fl, dr), (dz.numel() + fl.numel()) * 2 + dr.numel() * 4, 2.0 * B * HW * N * C)
            d_rows = dr.to(rows_dtype)
        return d_rows, d_feat, None, None


def mask_rows_loss_supported(rows, feat):
    """The fused contraction + loss statistics apply: bf16 channels_last CUDA feature map whose pixel count is a multiple of
    the 128-row tile and whose channel count is a multiple of 8."""
    if not (feat.is_cuda and feat.dim() == 4 and feat.dtype == torch.bfloat16 and rows.dim() == 3):
        return False
    B, C, H, W = feat.shape
    return (H * W) % 128 == 0 and C % 8 == 0 and feat.permute(0, 2, 3, 1).is_contiguous() and os.environ.get("DFINE_MASK_FUSED", "1") != "0"


def mask_rows_loss(rows, feat, tgt, alpha):
    """rows [B, N, C], feat [B, C, H, W] channels_last bf16, tgt [B, G, H*W] in [0, 1], alpha [B, G] -> stats [B, N, 3] fp32,
    differentiable w.r.t. rows and feat; == ``mask_row_stats(mask_logits_rows(rows, feat).flatten(2), tgt, alpha)`` without the
    logits tensor (reference dfine_decoder.py:937-940 + dfine_criterion.py:272-357)."""
    return _MaskRowsLoss.apply(rows, feat, tgt, alpha)


class _BilinearAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, base, x):
        _lib.require_gpu(x)
        lib = _lib.load()
        xl, bl = _nhwc(x), _nhwc(base)
        B, h, w, C = xl.shape
        _, H, W, _ = bl.shape
        y = torch.empty_like(bl)
        with _lib.timed("bilinear_add_fwd", (xl.numel() + 2 * bl.numel()) * xl.element_size()):
            rc = lib.dfine_bilinear_add_fwd(_lib.ptr(xl), _lib.ptr(bl), _lib.dtype_code(xl), _lib.ptr(y), B, h, w, H, W, C, _lib.stream_ptr())
        _lib.check(rc, "dfine_bilinear_add_fwd")
        ctx.meta = (B, h, w, H, W, C)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        B, h, w, H, W, C = ctx.meta
        lib = _lib.load()
        gyl = _nhwc(gy)
        gx = torch.empty(B, h, w, C, dtype=gyl.dtype, device=gyl.device)
        with _lib.timed("bilinear_bwd", (gx.numel() + gyl.numel()) * gx.element_size()):
            rc = lib.dfine_bilinear_bwd(_lib.ptr(gyl), _lib.dtype_code(gyl), _lib.ptr(gx), B, h, w, H, W, C, _lib.stream_ptr())
        _lib.check(rc, "dfine_bilinear_bwd")
        return gy, gx.permute(0, 3, 1, 2)


def bilinear_add(base, x):
    """``base + F.interpolate(x, size=base.shape[-2:], mode="bilinear", align_corners=False)`` (MaskPixelDecoder's top-down
    sum, dfine_decoder.py:352-360). One HIP launch forward, one gather launch backward for channels_last bf16 / fp32 maps on the
    GPU that are up-sampled by at most 8; ATen otherwise."""
    C = x.shape[1]
    ok = (x.is_cuda and x.dtype == base.dtype and x.dtype in (torch.float32, torch.bfloat16) and x.dim() == 4
          and base.shape[:2] == x.shape[:2] and C % (16 // x.element_size()) == 0
          and all(o >= i and -(-o // i) <= 8 for o, i in zip(base.shape[-2:], x.shape[-2:])))
    if not ok:
        return base + torch.nn.functional.interpolate(x, size=base.shape[-2:], mode="bilinear", align_corners=False)
    return _BilinearAdd.apply(base, x)


class _Upsample2x(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slot):
        _lib.require_gpu(x)
        lib = _lib.load()
        xl = _nhwc(x)
        B, H, W, C = xl.shape
        ldy = 0
        if slot is not None:
            y = slot[0].buf[..., slot[1] : slot[1] + C]
            ldy = slot[0].buf.shape[-1]
            if y.dtype != xl.dtype or tuple(y.shape) != (B, 2 * H, 2 * W, C):
                raise _lib.DfineHipError("upsample_nearest2x: the concat slot does not match the output")
        else:
            y = torch.empty(B, 2 * H, 2 * W, C, dtype=xl.dtype, device=xl.device)
        with _lib.timed("upsample2x_fwd", 5 * xl.numel() * xl.element_size()):
            rc = lib.dfine_upsample2x_fwd(_lib.ptr(xl), _lib.dtype_code(xl), _lib.ptr(y), B, H, W, C, ldy, _lib.stream_ptr())
        _lib.check(rc, "dfine_upsample2x_fwd")
        ctx.meta = (B, H, W, C, xl.dtype, slot)
        ctx.set_materialize_grads(False)
        return y.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        B, H, W, C, dtype, slot = ctx.meta
        lib = _lib.load()
        if slot is not None and slot[0].grad is not None:
            g2 = slot[0].grad[..., slot[1] : slot[1] + C].to(dtype)
            gy = g2 if gy is None else g2 + gy.permute(0, 2, 3, 1)
        elif gy is None:
            return None, None
        else:
            gy = gy.to(dtype).permute(0, 2, 3, 1)
        gyl, ld = _grad_rows(gy, C)
        gx = torch.empty(B, H, W, C, dtype=dtype, device=gyl.device)
        with _lib.timed("upsample2x_bwd", 5 * gx.numel() * gx.element_size()):
            rc = lib.dfine_upsample2x_bwd(_lib.ptr(gyl), ld, _lib.dtype_code(gx), _lib.ptr(gx), B, H, W, C, _lib.stream_ptr())
        _lib.check(rc, "dfine_upsample2x_bwd")
        return gx.permute(0, 3, 1, 2), None


def upsample_nearest2x(x, into=None):
    """``F.interpolate(x, scale_factor=2., mode="nearest")`` on a channels_last map (the FPN's upsampling,
    hybrid_encoder.py:377); ``into`` (a ConcatBuffer): the result becomes the buffer's next channel slice."""
    return _Upsample2x.apply(x, (into, into.take(x.shape[1])) if into is not None else None)


def upsample_nearest2x_supported(x):
    return x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.shape[1] % (8 if x.dtype == torch.bfloat16 else 4) == 0


class _GateMix(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pre, a, b):
        lib = _lib.load()
        D = a.shape[-1]
        pre, a, b = pre.contiguous(), a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        rc = lib.dfine_gate_mix_fwd(_lib.ptr(pre), _lib.ptr(a), _lib.ptr(b), _lib.dtype_code(a), _lib.ptr(out), a.numel() // D, D,
                                    _lib.stream_ptr())
        _lib.check(rc, "dfine_gate_mix_fwd")
        ctx.save_for_backward(pre, a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        pre, a, b = ctx.saved_tensors
        lib = _lib.load()
        D = a.shape[-1]
        g = g.to(a.dtype).contiguous()
        dpre, da, db = torch.empty_like(pre), torch.empty_like(a), torch.empty_like(b)
        rc = lib.dfine_gate_mix_bwd(_lib.ptr(pre), _lib.ptr(a), _lib.ptr(b), _lib.ptr(g), _lib.dtype_code(a), _lib.ptr(dpre), _lib.ptr(da),
                                    _lib.ptr(db), a.numel() // D, D, _lib.stream_ptr())
        _lib.check(rc, "dfine_gate_mix_bwd")
        return dpre, da, db


def gate_mix(pre, a, b):
    """``g1 * a + g2 * b`` with ``g1, g2 = sigmoid(pre).chunk(2, -1)`` (the decoder gateway between its Linear and its
    LayerNorm, dfine_decoder.py:181-196): one HIP launch forward, one backward, on the GPU for bf16 / fp32 of one dtype."""
    if (a.is_cuda and a.dtype in (torch.float32, torch.bfloat16) and pre.dtype == a.dtype == b.dtype
            and pre.shape[-1] == 2 * a.shape[-1] and a.shape == b.shape):
        return _GateMix.apply(pre, a, b)
    g1, g2 = torch.sigmoid(pre).chunk(2, -1)
    return g1 * a + g2 * b


MSDA_PREPARE_FUSED = os.environ.get("DFINE_MSDA_PREPARE", "1") != "0"  # measurement switch


class _MsdaPrepare(torch.autograd.Function):
    @staticmethod
    def forward(ctx, offsets, logits, ref_boxes, point_scale, n_head):
        lib = _lib.load()
        P = point_scale.numel()
        n_tok = ref_boxes.numel() // 4
        offsets, logits = offsets.contiguous(), logits.contiguous()
        ref = ref_boxes.detach().float().contiguous()
        scale = point_scale.float().contiguous()
        lead = ref_boxes.shape[:-1]
        loc = torch.empty(*lead, n_head, P, 2, dtype=torch.float32, device=offsets.device)
        weights = torch.empty(*lead, n_head, P, dtype=torch.float32, device=offsets.device)
        rc = lib.dfine_msda_prepare_fwd(_lib.ptr(offsets), _lib.ptr(logits), _lib.dtype_code(offsets), _lib.ptr(ref), _lib.ptr(scale),
                                        _lib.ptr(loc), _lib.ptr(weights), n_tok, n_head, P, _lib.stream_ptr())
        _lib.check(rc, "dfine_msda_prepare_fwd")
        ctx.save_for_backward(weights, ref, scale)
        ctx.meta = (offsets.shape, logits.shape, offsets.dtype, n_head, P, n_tok)
        return loc, weights

    @staticmethod
    def backward(ctx, g_loc, g_w):
        weights, ref, scale = ctx.saved_tensors
        off_shape, logit_shape, dtype, n_head, P, n_tok = ctx.meta
        lib = _lib.load()
        g_loc = torch.zeros_like(weights).unsqueeze(-1).expand(*weights.shape, 2).contiguous() if g_loc is None else g_loc.float().contiguous()
        g_w = torch.zeros_like(weights) if g_w is None else g_w.float().contiguous()
        d_off = torch.empty(off_shape, dtype=dtype, device=weights.device)
        d_logit = torch.empty(logit_shape, dtype=dtype, device=weights.device)
        rc = lib.dfine_msda_prepare_bwd(_lib.ptr(g_loc), _lib.ptr(g_w), _lib.ptr(weights), _lib.ptr(ref), _lib.ptr(scale),
                                        _lib.dtype_code(d_off), _lib.ptr(d_off), _lib.ptr(d_logit), n_tok, n_head, P, _lib.stream_ptr())
        _lib.check(rc, "dfine_msda_prepare_bwd")
        return d_off, d_logit, None, None, None


def msda_prepare(offsets, logits, ref_boxes, point_scale, n_head):
    """Sampling locations and attention weights of the deformable cross-attention from its two projections
    (dfine_decoder.py:148-166): ``loc = ref.xy + offsets * (point_scale * 0.5) * ref.wh`` as float32 [..., nH, P, 2] and
    ``softmax`` over the P points of each head as float32 [..., nH, P]. One HIP launch forward and one backward on the GPU when
    the reference boxes carry no gradient (the decoder detaches them); the op-by-op form otherwise."""
    P = point_scale.numel()
    lead = ref_boxes.shape[:-1]
    if (MSDA_PREPARE_FUSED and offsets.is_cuda and offsets.dtype in (torch.float32, torch.bfloat16) and offsets.dtype == logits.dtype
            and ref_boxes.shape[-1] == 4 and not (torch.is_grad_enabled() and ref_boxes.requires_grad)
            and offsets.numel() == ref_boxes.numel() // 4 * n_head * P * 2 and logits.numel() == ref_boxes.numel() // 4 * n_head * P):
        return _MsdaPrepare.apply(offsets, logits, ref_boxes, point_scale, n_head)
    off = offsets.float().view(*lead, n_head, P, 2)
    weights = torch.softmax(logits.float().view(*lead, n_head, P), dim=-1)
    ref = ref_boxes.float()[..., None, None, :]
    loc = ref[..., :2] + off * (point_scale.view(P, 1) * 0.5) * ref[..., 2:]
    return loc, weights


def row_max(x):
    """``x.max(-1).values`` as float32 for a dense bf16 / fp32 CUDA tensor, one HIP launch (query selection's best class
    logit per token); not differentiable. Anything else goes to ATen."""
    if not (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.is_contiguous()) or (torch.is_grad_enabled() and x.requires_grad):
        return x.max(-1).values
    lib = _lib.load()
    C = x.shape[-1]
    out = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    rc = lib.dfine_row_max(_lib.ptr(x), _lib.dtype_code(x), _lib.ptr(out), x.numel() // C, C, _lib.stream_ptr())
    _lib.check(rc, "dfine_row_max")
    return out


def stem_pool_supported(x):
    return x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.shape[1] % (8 if x.dtype == torch.bfloat16 else 4) == 0


class _AddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, weight, bias, eps, clamp):
        _lib.require_gpu(x, weight, bias)
        lib = _lib.load()
        D = x.shape[-1]
        xc = x.contiguous()
        rc_ = res.to(xc.dtype).contiguous() if res is not None else None
        M = xc.numel() // D
        y = torch.empty_like(xc)
        keep_s = res is not None or clamp > 0
        s = torch.empty_like(xc) if keep_s else xc
        mean = torch.empty(M, dtype=torch.float32, device=xc.device)
        rstd = torch.empty(M, dtype=torch.float32, device=xc.device)
        g32, b32 = weight.detach().float(), bias.detach().float()
        es = xc.element_size()
        with _lib.timed("add_layernorm_fwd", xc.numel() * es * (2 + (1 if res is not None else 0) + (1 if keep_s else 0))):
            rc = lib.dfine_add_layernorm_fwd(_lib.ptr(xc), _lib.ptr(rc_), _lib.dtype_code(xc), _lib.ptr(g32), _lib.ptr(b32), _lib.ptr(y),
                                             _lib.ptr(s) if keep_s else None, _lib.ptr(mean), _lib.ptr(rstd), M, D, eps, clamp,
                                             _lib.stream_ptr())
        _lib.check(rc, "dfine_add_layernorm_fwd")
        ctx.save_for_backward(s, g32, mean, rstd)
        ctx.meta = (weight, bias, clamp, res is not None, res.dtype if res is not None else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        s, g32, mean, rstd = ctx.saved_tensors
        weight, bias, clamp, has_res, res_dtype = ctx.meta
        lib = _lib.load()
        D = s.shape[-1]
        M = s.numel() // D
        gy = gy.to(s.dtype).contiguous()
        ds = torch.empty_like(s)
        direct = (DIRECT_WGRAD_ACCUM and weight.grad is not None and bias.grad is not None and weight.grad.dtype == torch.float32
                  and weight.grad.is_contiguous() and bias.grad.is_contiguous())
        if direct:
            dg, db = weight.grad, bias.grad
        else:
            dg, db = zeros_f32(D, s.device), zeros_f32(D, s.device)
        with _lib.timed("add_layernorm_bwd", s.numel() * s.element_size() * 3):
            rc = lib.dfine_add_layernorm_bwd(_lib.ptr(gy), _lib.ptr(s), _lib.dtype_code(s), _lib.ptr(g32), _lib.ptr(mean), _lib.ptr(rstd),
                                             _lib.ptr(ds), _lib.ptr(dg), _lib.ptr(db), M, D, clamp, _lib.stream_ptr())
        _lib.check(rc, "dfine_add_layernorm_bwd")
        d_res = (ds if res_dtype == ds.dtype else ds.to(res_dtype)) if has_res else None
        if direct:
            return ds, d_res, None, None, None, None
        return ds, d_res, dg.clone().to(weight.dtype), db.clone().to(bias.dtype), None, None


def add_layer_norm_supported(x, dim):
    return x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and (dim in (128, 256, 384) or (dim == 512 and x.dtype == torch.bfloat16))


def add_layer_norm(x, res, weight, bias, eps=1e-5, clamp=None):
    """``F.layer_norm(clamp(x + res), (D,), weight, bias, eps)`` in x's dtype (statistics in fp32), one launch
    forward and one backward; ``res`` / ``clamp`` optional. Reference: the post-norm blocks of
    dfine_decoder.py:233-271 and hybrid_encoder.py:274-290."""
    return _AddLayerNorm.apply(x, res, weight, bias, float(eps), float(clamp) if clamp else 0.0)


class _MaskRowStats(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, tgt, alpha):
        _lib.require_gpu(logits, tgt, alpha)
        lib = _lib.load()
        B, M, HW = logits.shape
        G = tgt.shape[1]
        z = logits.contiguous()
        t = tgt.float().contiguous()
        a = alpha.float().contiguous()
        stats = zeros_f32(B * M * 3, z.device).view(B, M, 3)
        with _lib.timed("mask_loss_stats", z.numel() * z.element_size() + t.numel() * 4 * (M // G)):
            rc = lib.dfine_mask_loss_stats(_lib.ptr(z), _lib.dtype_code(z), _lib.ptr(t), _lib.ptr(a), _lib.ptr(stats),
                                           B, M, G, HW, _lib.stream_ptr())
        _lib.check(rc, "dfine_mask_loss_stats")
        ctx.save_for_backward(z, t, a)
        return stats

    @staticmethod
    def backward(ctx, g_stats):
        z, t, a = ctx.saved_tensors
        lib = _lib.load()
        B, M, HW = z.shape
        G = t.shape[1]
        g = g_stats.float().contiguous()
        dz = torch.empty_like(z)
        with _lib.timed("mask_loss_grad", 2 * z.numel() * z.element_size() + t.numel() * 4 * (M // G)):
            rc = lib.dfine_mask_loss_grad(_lib.ptr(z), _lib.dtype_code(z), _lib.ptr(t), _lib.ptr(a), _lib.ptr(g), _lib.ptr(dz),
                                          B, M, G, HW, _lib.stream_ptr())
        _lib.check(rc, "dfine_mask_loss_grad")
        return dz, None, None


def mask_row_stats(logits, tgt, alpha):
    """logits [B, M, HW] (f32/bf16) of the matched queries, tgt [B, G, HW] in [0,1] (M a multiple of G:
    slot m is ground truth m % G), alpha [B, G] -> stats [B, M, 3] fp32 = (sum of the focal-BCE terms,
    sum p*t, sum p) per row, differentiable w.r.t. logits (dfine_criterion.py:272-357)."""
    return _MaskRowStats.apply(logits, tgt, alpha)


class _FdrHead(torch.autograd.Function):
    """softmax -> Integral -> distance2bbox and the LQE statistics in one launch (csrc/fdr.hip)."""

    @staticmethod
    def forward(ctx, corners, ref_boxes, W, reg_scale, k):
        _lib.require_gpu(corners, ref_boxes, W)
        lib = _lib.load()
        lead = corners.shape[:-1]
        n_bins = W.numel()
        c = corners.float().contiguous().view(-1, 4, n_bins)
        r = ref_boxes.float().contiguous().view(-1, 4)
        Wc = W.float().contiguous()
        rows = c.shape[0]
        boxes = torch.empty(rows, 4, dtype=torch.float32, device=c.device)
        stat = torch.empty(rows, 4, k + 1, dtype=torch.float32, device=c.device)
        top = torch.empty(rows, 4, k, dtype=torch.uint8, device=c.device)
        with _lib.timed("fdr_head_fwd", c.numel() * 4 + rows * (16 + 16 + 4 * (k + 1) * 4)):
            rc = lib.dfine_fdr_head_fwd(_lib.ptr(c), _lib.ptr(r), _lib.ptr(Wc), _lib.ptr(boxes), _lib.ptr(stat), _lib.ptr(top),
                                        rows, n_bins, k, reg_scale, _lib.stream_ptr())
        _lib.check(rc, "dfine_fdr_head_fwd")
        ctx.save_for_backward(c, r, Wc, top)
        ctx.meta = (reg_scale, k, tuple(corners.shape))
        return boxes.view(*lead, 4), stat.view(*lead, 4 * (k + 1))

    @staticmethod
    def backward(ctx, g_boxes, g_stat):
        c, r, Wc, top = ctx.saved_tensors
        reg_scale, k, shape = ctx.meta
        lib = _lib.load()
        rows, _, n_bins = c.shape
        gb = g_boxes.float().contiguous() if g_boxes is not None else None
        gs = g_stat.float().contiguous() if g_stat is not None else None
        g_corners = torch.empty_like(c)
        with _lib.timed("fdr_head_bwd", c.numel() * 8 + rows * (16 + 16 + 4 * (k + 1) * 4)):
            rc = lib.dfine_fdr_head_bwd(_lib.ptr(c), _lib.ptr(r), _lib.ptr(Wc), _lib.ptr(top), _lib.ptr(gb), _lib.ptr(gs),
                                        _lib.ptr(g_corners), rows, n_bins, k, reg_scale, _lib.stream_ptr())
        _lib.check(rc, "dfine_fdr_head_bwd")
        return g_corners.view(shape), None, None, None, None


def fdr_head(corners, ref_boxes, W, reg_scale, k=4):
    """corners [..., 4*(reg_max+1)] (logits over the W(n) bins per box side), ref_boxes [..., 4] cxcywh
    (no gradient), W [reg_max+1] -> (boxes [..., 4] cxcywh fp32, stat [..., 4*(k+1)] fp32): the
    reference's ``distance2bbox(ref, Integral(corners), reg_scale)`` and the LQE input
    ``cat(prob.topk(k), mean)`` (dfine_decoder.py:276-313, 487-509), differentiable w.r.t. corners."""
    return _FdrHead.apply(corners, ref_boxes, W, float(reg_scale), int(k))


# Weight-gradient kernels on a SIDE stream. In backward a layer's weight gradient and its input gradient both consume dY and
# are independent of each other; the weight gradients are needed only by the optimizer at the end of the step. With a side
# stream installed (``TrainStep`` does) every wgrad launch forks from the current stream, runs beside the dgrad / BatchNorm
# chain of the layers below, and the step joins once before the optimizer (``join_wgrad``). Under graph capture the fork /
# join become graph edges, so the replayed graph has two parallel branches: the tails of the ~1 000 small backward kernels
# (most of which do not fill 256 CUs) overlap. Operands are ``record_stream``-ed so the allocator does not recycle them early.
wgrad_stream = None


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class on_wgrad_stream:
    def __init__(self, *tensors):
        self.tensors = tensors

    def __enter__(self):
        self.ctx = None
        if wgrad_stream is not None:
            wgrad_stream.wait_stream(torch.cuda.current_stream())
            for t in self.tensors:
                if t is not None:
                    t.record_stream(wgrad_stream)
            self.ctx = torch.cuda.stream(wgrad_stream)
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def join_wgrad():
    """Launch the queued (deferred) weight gradients; the current stream waits for every weight-gradient launch forked to the
    side stream. After this the fp32 gradients of every layer whose backward has run are complete (in stream order)."""
    if wgrad_queue is not None:
        wgrad_queue.flush()
    if wgrad_stream is not None:
        torch.cuda.current_stream().wait_stream(wgrad_stream)


class WgradQueue:
    """Deferred, grouped weight gradients (include/dfine_hip.h: dfine_wgrad_group_*). While a queue is installed
    (``ops.wgrad_queue``; ``TrainStep`` does) the backward of every Linear / dense Conv2d that accumulates straight into the
    flat fp32 gradient only RECORDS its problem -- (dY, X) stay alive, which costs a few GB of the 288 -- and ``flush()``
    launches them all at once: the ~190 weight gradients of a D-FINE-m step, most of which cannot fill 256 CUs on their own and
    pay a partial-tile atomic tail each, become one full-chip launch per tile shape with long row chunks. Problems the aligned
    kernel cannot serve (the 3-channel stem, N % 8 != 0 heads) are launched on the spot by the caller as before.

    Under hipGraph capture the launch table lives in PERSISTENT device memory reserved before the capture (``reserve``) and
    is uploaded after it (``upload_pending``): a captured launch only records pointers. The table must not come from the
    graph's own memory pool -- a block allocated late in a capture is one that earlier kernels of the same graph used and
    freed, and every replay would run those kernels over the table before the grouped launch reads it."""

    def __init__(self):
        self.problems, self.keep = [], []
        self.bytes = self.flops = 0
        self.pending_uploads = []
        self.table_chunks, self.table_off = [], 0  # persistent memory for the tables of captured launches
        self.max_need = 0
        self.launches = self.single_fallbacks = 0

    def reserve(self, device):
        """Outside any capture: make room for the tables of the next captured step (a few launches of the largest table seen)."""
        want = max(4 << 20, 4 * self.max_need)
        if not self.table_chunks or self.table_chunks[-1].numel() - self.table_off < want:
            self.table_chunks.append(torch.empty(want, dtype=torch.uint8, device=device))
            self.table_off = 0

    def _persistent(self, need):
        if not self.table_chunks or self.table_chunks[-1].numel() - self.table_off < need:
            return None
        view = self.table_chunks[-1][self.table_off : self.table_off + need]
        self.table_off += (need + 255) // 256 * 256
        return view

    def _launch_singly(self):
        lib = _lib.load()
        for q in self.problems:
            if q.conv:
                rc = lib.dfine_conv_wgrad(q.dY, q.X, q.dW, q.B, q.H, q.W, q.CI, q.OH, q.OW, q.CO, q.KH, q.KW, q.stride, q.pad, q.ldy, q.ldx,
                                          _lib.stream_ptr())
            else:
                rc = lib.dfine_linear_wgrad(q.dY, q.X, q.dW, q.db, q.M, q.N, q.K, q.ldw, q.ldy, q.ldx, _lib.stream_ptr())
            _lib.check(rc, "queued wgrad")
        self.single_fallbacks += 1
        self.discard()

    def add(self, problem, keep, nbytes, flops):
        """Queue ``problem`` (a _lib.WgradProblem) if the grouped kernel can serve it; False: launch it yourself."""
        lib = _lib.load()
        if not lib.dfine_wgrad_group_eligible(ctypes.byref(problem)):
            return False
        self.problems.append(problem)
        self.keep.append(keep)
        self.bytes += nbytes
        self.flops += flops
        return True

    def flush(self):
        n = len(self.problems)
        if n == 0:
            return
        lib = _lib.load()
        if os.environ.get("DFINE_WGRAD_QUEUE_SINGLE") == "1":  # diagnostic: deferred, but one launch per problem
            return self._launch_singly()
        arr = (_lib.WgradProblem * n)(*self.problems)
        need = lib.dfine_wgrad_group_plan(arr, n, None, 0)
        if need <= 0:
            _lib.check(int(need) if need < 0 else -1, "dfine_wgrad_group_plan")
        device = self.keep[0][0].device
        capturing = torch.cuda.is_current_stream_capturing()
        host = torch.empty(int(need), dtype=torch.uint8, pin_memory=not capturing)
        used = lib.dfine_wgrad_group_plan(arr, n, ctypes.c_void_p(host.data_ptr()), int(need))
        if used != need:
            _lib.check(int(used) if used < 0 else -1, "dfine_wgrad_group_plan")
        self.max_need = max(self.max_need, int(need))
        if capturing:
            dev = self._persistent(int(need))
            if dev is None:  # nobody reserved table memory before this capture: correct, just not grouped
                return self._launch_singly()
            self.pending_uploads.append((dev, host))
        else:
            dev = torch.empty(int(need), dtype=torch.uint8, device=device)
            dev.copy_(host, non_blocking=True)
        with _lib.timed("wgrad", self.bytes, f"grouped x{n}", flops=self.flops):
            rc = lib.dfine_wgrad_group_launch(ctypes.c_void_p(host.data_ptr()), _lib.ptr(dev), _lib.stream_ptr())
        _lib.check(rc, "dfine_wgrad_group_launch")
        self.launches += 1
        self.problems, self.keep = [], []
        self.bytes = self.flops = 0

    def upload_pending(self):
        """After a capture ended: fill the device tables its grouped launches will read on replay."""
        for dev, host in self.pending_uploads:
            dev.copy_(host)
        self.pending_uploads = []

    def discard(self):
        """Drop queued problems without launching (a failed / abandoned backward)."""
        self.problems, self.keep = [], []
        self.bytes = self.flops = 0


wgrad_queue = None


# Accumulate Linear weight gradients straight into ``param.grad`` (the flat fp32 gradient buffer) and
# hand autograd ``None`` for them -- the same contract as a fused wgrad-accumulate GEMM. Switched off
# by the data-parallel reducer, which counts post-accumulate hooks to know when a bucket is complete.
DIRECT_WGRAD_ACCUM = True


def bf16_of(param):
    """bf16 value of an fp32 master parameter: the optimizer-maintained shadow copy when there is one and
    it is current (optim.FlatState.enable_bf16_shadow), else a cast."""
    shadow = getattr(param, "_bf16_shadow", None)
    if shadow is not None and param._shadow_version == param._version:
        return shadow
    return param.to(torch.bfloat16)


def transpose_weight(w16):
    """bf16 weight with memory [K][T][C] (Linear [N, K_in]: T = 1; channels_last conv [CO, CI, KH, KW]) -> dense
    [C][T][K] (Linear: W^T; conv: [CI, KH, KW, CO]) -- the operand of the input-gradient GEMM."""
    if w16.dim() == 2:
        return w16.t().contiguous()
    return w16.permute(1, 2, 3, 0).contiguous()


def bf16_t_of(param):
    """Transposed bf16 copy of a weight (see ``transpose_weight``): the optimizer-maintained transposed shadow when
    there is one and it is current, else computed here."""
    st = getattr(param, "_bf16_shadow_t", None)
    if st is not None and param._shadow_version == param._version:
        return st
    return transpose_weight(bf16_of(param))


def _khwc(w16):
    """[CO, CI, KH, KW] (channels_last memory) -> dense [CO, KH, KW, CI] view (a copy only if it was not channels_last)."""
    v = w16.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


class _Linear(torch.autograd.Function):
    """y = relu?(x @ W[rows]^T + b[rows]) in bf16: forward and input gradient by the implicit-GEMM MFMA kernel
    (dfine_igemm_bf16), weight/bias gradient by the split-M MFMA kernel (dfine_linear_wgrad)."""

    @staticmethod
    def forward(ctx, x, weight, bias, row_begin, row_end, relu, relu_in=False, mask_by_consumer=False):
        from . import gemm

        x16 = x.to(torch.bfloat16)
        K = weight.shape[1]
        x2 = x16.reshape(-1, K)
        w16 = bf16_of(weight)[row_begin:row_end]
        b32 = bias.detach()[row_begin:row_end].float() if bias is not None else None
        y = gemm.linear_forward(x2, w16, b32, relu)
        ctx.save_for_backward(x2, y if (relu and not mask_by_consumer) else None)
        # relu_in: x is the output of the previous Linear's fused ReLU and THIS layer's input-gradient GEMM applies that ReLU's
        # backward in its epilogue; mask_by_consumer: this layer's own ReLU is handled that way by the next layer (MLP chains)
        ctx.relu_chain = (bool(relu_in), bool(mask_by_consumer))
        ctx.params = (weight, bias, row_begin, row_end, relu, x16.shape)
        return y.reshape(*x16.shape[:-1], row_end - row_begin) if y.is_contiguous() else y.unflatten(0, x16.shape[:-1])

    @staticmethod
    def backward(ctx, gy):
        from . import gemm

        x2, y = ctx.saved_tensors
        weight, bias, r0, r1, relu, x_shape = ctx.params
        lib = _lib.load()
        N, K = r1 - r0, weight.shape[1]
        relu_in, mask_by_consumer = ctx.relu_chain
        gy16 = gy.to(torch.bfloat16).reshape(-1, N)
        if relu and not mask_by_consumer:  # (otherwise the gradient arrives masked from the next layer's input-gradient GEMM)
            gy16 = torch.ops.aten.threshold_backward(gy16, y.reshape(-1, N), 0)  # one launch instead of compare + multiply
        if not gy16.is_contiguous():
            gy16 = gy16.contiguous()
        M = x2.shape[0]
        dx = None
        if ctx.needs_input_grad[0]:
            wt = bf16_t_of(weight)[:, r0:r1]  # [K, N] rows of pitch N_total: read in place
            dx = gemm.linear_forward(gy16, wt, None, False, name="gemm_dgrad", relu_mask=x2 if relu_in else None).view(x_shape)
        direct = (DIRECT_WGRAD_ACCUM and weight.grad is not None and weight.grad.dtype == torch.float32
                  and weight.grad.is_contiguous() and (bias is None or (bias.grad is not None and bias.grad.is_contiguous())))
        if direct:
            dw, db = weight.grad[r0:r1], (bias.grad[r0:r1] if bias is not None else None)
        else:
            dw = torch.zeros(N, K, dtype=torch.float32, device=x2.device)
            db = torch.zeros(N, dtype=torch.float32, device=x2.device) if bias is not None else None
        nbytes, flops = (M * N + M * K) * 2 + N * K * 4, 2.0 * M * N * K
        if direct and wgrad_queue is not None:
            prob = _lib.WgradProblem(dY=gy16.data_ptr(), X=x2.data_ptr(), dW=dw.data_ptr(), db=db.data_ptr() if db is not None else None,
                                     conv=0, M=M, N=N, K=K, ldw=K, ldy=N, ldx=K)
            if wgrad_queue.add(prob, (gy16, x2), nbytes, flops):
                return dx, None, None, None, None, None, None, None
        with on_wgrad_stream(gy16, x2 if direct else None) if direct else _NullCtx():
            with _lib.timed("wgrad", nbytes, f"linear M{M} N{N} K{K}", flops=flops):
                rc = lib.dfine_linear_wgrad(_lib.ptr(gy16), _lib.ptr(x2), _lib.ptr(dw), _lib.ptr(db), M, N, K, K, N, K, _lib.stream_ptr())
        _lib.check(rc, "dfine_linear_wgrad")
        if direct:
            return dx, None, None, None, None, None, None, None
        full_w = dw if (r0 == 0 and r1 == weight.shape[0]) else torch.zeros_like(weight, dtype=torch.float32).index_put_((torch.arange(r0, r1, device=dw.device),), dw)
        full_b = None
        if bias is not None:
            full_b = db if (r0 == 0 and r1 == bias.shape[0]) else torch.zeros_like(bias, dtype=torch.float32).index_put_((torch.arange(r0, r1, device=db.device),), db)
        return dx, full_w.to(weight.dtype), (full_b.to(bias.dtype) if full_b is not None else None), None, None, None, None, None


# MLP chains: a hidden ReLU's backward applied by the next layer's input-gradient GEMM (dfine_igemm_bf16_relumask) instead of a
# threshold_backward launch per layer; DFINE_RELU_EPILOGUE=0 switches back (A/B measurements)
RELU_IN_NEXT_EPILOGUE = os.environ.get("DFINE_RELU_EPILOGUE", "1") != "0"


def linear(x, weight, bias=None, row_begin=0, row_end=None, relu=False, relu_in=False, mask_by_consumer=False):
    """``relu?(F.linear(x, weight[row_begin:row_end], bias[row_begin:row_end]))``. On the GPU in bf16 mode (autocast, or
    bf16 input) forward, input gradient and weight/bias gradient are all HIP MFMA kernels; otherwise (fp32 parity
    runs, inference without grad on odd shapes) it is plain ``F.linear``."""
    row_end = weight.shape[0] if row_end is None else row_end
    n, k = row_end - row_begin, weight.shape[1]
    bf16_mode = x.is_cuda and (torch.is_autocast_enabled() or x.dtype == torch.bfloat16)
    if not bf16_mode:
        w = weight if (row_begin == 0 and row_end == weight.shape[0]) else weight[row_begin:row_end]
        b = bias if (bias is None or (row_begin == 0 and row_end == weight.shape[0])) else bias[row_begin:row_end]
        y = torch.nn.functional.linear(x, w, b)
        return torch.relu(y) if relu else y
    if not (torch.is_grad_enabled() and (weight.requires_grad or x.requires_grad)):
        from . import gemm

        x16 = x.to(torch.bfloat16)
        b32 = bias.detach()[row_begin:row_end].float() if bias is not None else None
        y = gemm.linear_forward(x16.reshape(-1, k), bf16_of(weight)[row_begin:row_end], b32, relu)
        return y.reshape(*x16.shape[:-1], n) if y.is_contiguous() else y.unflatten(0, x16.shape[:-1])
    if not RELU_IN_NEXT_EPILOGUE:
        relu_in = mask_by_consumer = False
    return _Linear.apply(x, weight, bias, row_begin, row_end, bool(relu), bool(relu_in), bool(mask_by_consumer))


class _DenseConv(torch.autograd.Function):
    """Dense (groups = 1, no bias) convolution on a channels_last map, all three passes on MFMA kernels of this
    library: forward and input gradient by the implicit-GEMM kernel (dfine_igemm_bf16; the forward can leave the
    following BatchNorm's statistics in ``stats`` and write into a channel slice ``out``), weight gradient by the
    split-M kernel with gathered rows, accumulated straight into the fp32 .grad."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding, pad_br, stats, out, bn_handle=None):
        from . import gemm

        ctx.bn_handle = bn_handle if stride == 1 else None
        if bn_handle is not None:
            bn_handle.consumers += 1
        x16 = x.to(torch.bfloat16)
        xl = gemm.nhwc(x16)
        if gemm.pixel_pitch(xl) is None:
            xl = xl.contiguous()
        wk = _khwc(bf16_of(weight))
        stats64 = stats if stats is not None else None
        yl = gemm.conv_forward(xl, wk, stride, padding, pad_br, out=gemm.nhwc(out) if out is not None else None, stats=stats64)
        ctx.save_for_backward(xl)
        ctx.weight, ctx.geom = weight, (stride, padding, pad_br)
        if out is not None:
            ctx.mark_dirty(out)
            return out
        return yl.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        from . import gemm

        (xl,) = ctx.saved_tensors
        weight = ctx.weight
        stride, padding, pad_br = ctx.geom
        lib = _lib.load()
        K_out, C_in, KH, KW = weight.shape
        gyl = gemm.nhwc(gy.to(torch.bfloat16))
        if gemm.pixel_pitch(gyl) is None:
            gyl = gyl.contiguous()
        B, H, W, _ = xl.shape
        OH, OW = gyl.shape[1], gyl.shape[2]
        dx = None
        if ctx.needs_input_grad[0]:
            h = ctx.bn_handle
            br = h.begin((H, W)) if h is not None else None
            dxl = gemm.conv_input_grad(gyl, bf16_t_of(weight), (H, W), stride, padding, bnred=br)
            if br is not None:
                h.finish(dxl, dxl._bnred_done)
            dx = dxl.permute(0, 3, 1, 2)
        dw = _conv_weight_grad(weight, gyl, xl, stride, padding)
        return dx, dw, None, None, None, None, None, None


def _conv_weight_grad(weight, gyc, xc, stride, padding):
    """Weight gradient of a dense conv from its output gradient ``gyc`` and input ``xc`` (NHWC pixel rows, possibly channel
    slices read in place): accumulated straight into the fp32 ``weight.grad`` -- queued for the grouped launch when a
    WgradQueue is installed -- and ``None`` returned; or computed into a fresh tensor and returned for autograd."""
    from . import gemm

    lib = _lib.load()
    K_out, C_in, KH, KW = weight.shape
    B, H, W, _ = xc.shape
    OH, OW = gyc.shape[1], gyc.shape[2]
    ldy, ldx = gemm.pixel_pitch(gyc), gemm.pixel_pitch(xc)
    taps = KH * KW
    # the fp32 gradient must be laid out [K_out][KH][KW][C_in] (a channels_last weight's memory order)
    # (is_leaf first: the stem's zero-padded weight is a non-leaf view, and asking one for .grad warns)
    direct = (DIRECT_WGRAD_ACCUM and weight.is_leaf and weight.grad is not None and weight.grad.dtype == torch.float32
              and _is_khwc(weight.grad))
    if direct:
        dw = weight.grad
    else:
        dw = torch.zeros(K_out, C_in, KH, KW, dtype=torch.float32, device=xc.device).contiguous(memory_format=torch.channels_last)
    M = B * OH * OW
    nbytes, flops = (M * K_out + xc.numel()) * 2 + K_out * C_in * taps * 4, 2.0 * M * K_out * C_in * taps
    if direct and wgrad_queue is not None:
        prob = _lib.WgradProblem(dY=gyc.data_ptr(), X=xc.data_ptr(), dW=dw.data_ptr(), db=None, conv=1, B=B, H=H, W=W, CI=C_in,
                                 OH=OH, OW=OW, CO=K_out, KH=KH, KW=KW, stride=stride, pad=padding, ldy=ldy, ldx=ldx)
        if wgrad_queue.add(prob, (gyc, xc), nbytes, flops):
            return None
    with on_wgrad_stream(gyc, xc) if direct else _NullCtx():
        with _lib.timed("wgrad", nbytes,
                        f"conv {B}x{H}x{W} {C_in}->{K_out} k{KH}s{stride}", flops=flops):
            rc = lib.dfine_conv_wgrad(_lib.ptr(gyc), _lib.ptr(xc), _lib.ptr(dw), B, H, W, C_in, OH, OW, K_out, KH, KW,
                                      stride, padding, ldy, ldx, _lib.stream_ptr())
    _lib.check(rc, "dfine_conv_wgrad")
    return None if direct else dw.to(weight.dtype)


class _DenseConvPair(torch.autograd.Function):
    """Two stride-1 dense convolutions of the SAME input (RepVgg's 3x3 + 1x1 branches, hybrid_encoder.py:106-121; CSPLayer's
    conv1 / conv2, :209-239) as one autograd node: forward = two implicit GEMMs (each may leave its BatchNorm statistics),
    backward = ONE input gradient -- the second GEMM adds into the first's output in its epilogue (dfine_igemm_bf16, act | 0x400)
    instead of autograd summing two maps with an extra kernel -- and two (queued) weight gradients."""

    @staticmethod
    def forward(ctx, x, wa, wb, pad_a, pad_b, stats_a, stats_b, bn_handle=None):
        from . import gemm

        ctx.bn_handle = bn_handle
        if bn_handle is not None:
            bn_handle.consumers += 1
        xl = gemm.nhwc(x.to(torch.bfloat16))
        if gemm.pixel_pitch(xl) is None:
            xl = xl.contiguous()
        ya = gemm.conv_forward(xl, _khwc(bf16_of(wa)), 1, pad_a, None, stats=stats_a)
        yb = gemm.conv_forward(xl, _khwc(bf16_of(wb)), 1, pad_b, None, stats=stats_b)
        ctx.save_for_backward(xl)
        ctx.meta = (wa, wb, pad_a, pad_b)
        ctx.set_materialize_grads(False)
        return ya.permute(0, 3, 1, 2), yb.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, ga, gb):
        from . import gemm

        (xl,) = ctx.saved_tensors
        wa, wb, pad_a, pad_b = ctx.meta
        B, H, W, _ = xl.shape

        def rows(g):
            gl = gemm.nhwc(g.to(torch.bfloat16))
            return gl if gemm.pixel_pitch(gl) is not None else gl.contiguous()

        dx, dwa, dwb = None, None, None
        live = [(g, w, pad) for g, w, pad in ((ga, wa, pad_a), (gb, wb, pad_b)) if g is not None]
        for i, (g, w, pad) in enumerate(live):
            gl = rows(g)
            if ctx.needs_input_grad[0]:
                # the launch that FINISHES the input gradient takes the producing norm's backward sums (BnBackwardHandle)
                h = ctx.bn_handle if i == len(live) - 1 else None
                br = h.begin((H, W)) if h is not None else None
                if dx is None:
                    dx = gemm.conv_input_grad(gl, bf16_t_of(w), (H, W), 1, pad, bnred=br)
                else:
                    gemm.conv_input_grad(gl, bf16_t_of(w), (H, W), 1, pad, out=dx, accumulate=True, bnred=br)
                if br is not None:
                    h.finish(dx, dx._bnred_done)
            dw = _conv_weight_grad(w, gl, xl, 1, pad)
            if w is wa:
                dwa = dw
            else:
                dwb = dw
        return (dx.permute(0, 3, 1, 2) if dx is not None else None), dwa, dwb, None, None, None, None, None


def dense_conv_pair(x, weight_a, weight_b, pad_a, pad_b, stats_a=None, stats_b=None):
    """(F.conv2d(x, weight_a, padding=pad_a), F.conv2d(x, weight_b, padding=pad_b)) for two stride-1 dense convs of one input,
    bf16 GPU path (see _DenseConvPair). ``stats_*``: as in ``dense_conv``."""
    return _DenseConvPair.apply(x, weight_a, weight_b, int(pad_a), int(pad_b), stats_a, stats_b, getattr(x, "_bn_handle", None))


def _is_khwc(t):
    """True when the 4-d weight-shaped tensor ``t`` [K, C, KH, KW] is dense in [K][KH][KW][C] order."""
    K, C, KH, KW = t.shape
    want = (KH * KW * C, 1, KW * C, C)
    return all(t.shape[d] == 1 or t.stride(d) == want[d] for d in range(4))


def dense_conv(x, weight, stride=1, padding=0, pad_br=None, stats=None, out=None):
    """``F.conv2d(x, weight, None, stride, padding)`` for a groups=1 conv without bias on a channels_last map
    (``pad_br``: bottom/right padding when it differs from ``padding``, which then pads top/left only). In bf16 mode on
    the GPU all three passes are HIP MFMA kernels; ``stats`` (fp32 view of the fp64 replicas, ``bn_workspace``) receives
    the output's per-channel sums for the BatchNorm that follows; ``out`` (an NCHW-shaped channels_last view, e.g. a
    channel slice of a concat buffer) receives the result in place. Otherwise this is plain ``F.conv2d``."""
    bf16_mode = x.is_cuda and (torch.is_autocast_enabled() or x.dtype == torch.bfloat16)
    if not bf16_mode:
        if stats is not None or out is not None:
            raise _lib.DfineHipError("dense_conv: stats/out need the bf16 GPU path")
        if pad_br is not None and pad_br != padding:
            x = torch.nn.functional.pad(x, (padding, pad_br, padding, pad_br))
            padding = 0
        return torch.nn.functional.conv2d(x, weight, None, stride, padding)
    if not (torch.is_grad_enabled() and (weight.requires_grad or x.requires_grad)):
        from . import gemm

        xl = gemm.nhwc(x.to(torch.bfloat16))
        yl = gemm.conv_forward(xl, _khwc(bf16_of(weight)), stride, padding, pad_br, out=gemm.nhwc(out) if out is not None else None,
                               stats=stats)
        return out if out is not None else yl.permute(0, 3, 1, 2)
    return _DenseConv.apply(x, weight, int(stride), int(padding), pad_br, stats, out, getattr(x, "_bn_handle", None))


def conv_bias_act(x, conv, act=None, addend=None):
    """``act(conv(x) + bias + addend)`` for a deployed (norm-folded) nn.Conv2d, inference only: in bf16 mode on the GPU the
    dense conv runs on the implicit-GEMM kernel with the bias (and ReLU) in its epilogue, a depthwise conv on the depthwise
    kernel; otherwise the stock op."""
    bf16_mode = x.is_cuda and (torch.is_autocast_enabled() or x.dtype == torch.bfloat16) and not torch.is_grad_enabled()
    dense = conv.groups == 1 and conv.stride[0] == conv.stride[1] and conv.padding[0] == conv.padding[1] and conv.dilation == (1, 1)
    depthwise = conv.groups == conv.in_channels == conv.out_channels and conv.groups > 1
    if bf16_mode and dense:
        from . import gemm

        fuse_relu = act == "relu" and addend is None
        yl = gemm.conv_forward(gemm.nhwc(x.to(torch.bfloat16)), _khwc(conv.weight.detach().to(torch.bfloat16)), conv.stride[0],
                               conv.padding[0], bias=conv.bias.detach().float(), relu=fuse_relu)
        y = yl.permute(0, 3, 1, 2)
        if fuse_relu:
            return y
    elif bf16_mode and depthwise:
        y = depthwise_conv2d(x.to(torch.bfloat16), conv.weight, conv.stride[0]) + conv.bias.detach().to(torch.bfloat16).view(1, -1, 1, 1)
    else:
        y = conv(x)
    if addend is not None:
        y = y + addend
    if act == "relu":
        return torch.relu(y)
    if act == "silu":
        return torch.nn.functional.silu(y)
    return y


def attention_mask_bits(blocked):
    """bool [L, L] (True = may not attend) -> (bits_q, bits_k), int64 [L, ceil(L / 64)] each: the mask packed into 64-bit
    words along the key axis (bits_q[q][t] bit j = blocked[q][64 t + j]) and along the query axis (bits_k[k][t] bit j =
    blocked[64 t + j][k]) -- what dfine_attention_fwd / _bwd read instead of single bytes. A mask that is reused (the
    decoder's denoising mask, cached per GT count) keeps the pair in its ``_attn_bits`` attribute."""
    L = blocked.shape[0]
    n_t = (L + 63) // 64
    pad = torch.zeros(n_t * 64, n_t * 64, dtype=torch.bool, device=blocked.device)
    pad[:L, :L] = blocked
    weights = torch.ones(64, dtype=torch.int64, device=blocked.device) << torch.arange(64, device=blocked.device)  # bit 63 wraps: fine
    pack = lambda m: (m.view(n_t * 64, n_t, 64).to(torch.int64) * weights).sum(-1)[:L].contiguous()
    return pack(pad), pack(pad.t().contiguous())


class _SelfAttention(torch.autograd.Function):
    """softmax(scale * Q K^T + mask) V on the projections' own layout; see include/dfine_hip.h (dfine_attention_*)."""

    @staticmethod
    def forward(ctx, qk, v, blocked, n_head):
        _lib.require_gpu(qk, v)
        lib = _lib.load()
        B, L, D2 = qk.shape
        D = D2 // 2
        dh = D // n_head
        qk, v = qk.contiguous(), v.contiguous()
        k_view = qk[..., D:]
        mask = blocked.view(torch.uint8).contiguous() if blocked is not None else None
        bits_q = bits_k = None
        if blocked is not None:
            bits = getattr(blocked, "_attn_bits", None)
            bits_q, bits_k = bits if bits is not None else attention_mask_bits(blocked)
        o = torch.empty(B, L, D, dtype=torch.bfloat16, device=qk.device)
        lse = torch.empty(B, n_head, L, dtype=torch.float32, device=qk.device)
        scale = 1.0 / dh ** 0.5
        with _lib.timed("attention_fwd", (3 * B * L * D + B * L * D) * 2, f"B{B} L{L} H{n_head} dh{dh}"):
            rc = lib.dfine_attention_fwd(_lib.ptr(qk), _lib.ptr(k_view), _lib.ptr(v), D2, D2, D, _lib.ptr(mask), _lib.ptr(bits_q),
                                         _lib.ptr(o), D, _lib.ptr(lse), B, n_head, L, dh, scale, _lib.stream_ptr())
        _lib.check(rc, "dfine_attention_fwd")
        ctx.save_for_backward(qk, v, o, lse, mask, bits_q, bits_k)
        ctx.meta = (n_head, scale)
        return o

    @staticmethod
    def backward(ctx, go):
        qk, v, o, lse, mask, bits_q, bits_k = ctx.saved_tensors
        n_head, scale = ctx.meta
        lib = _lib.load()
        B, L, D2 = qk.shape
        D = D2 // 2
        dh = D // n_head
        go = go.to(torch.bfloat16).contiguous()
        dqk = torch.empty_like(qk)
        dv = torch.empty_like(v)
        dsum = torch.empty_like(lse)
        with _lib.timed("attention_bwd", 8 * B * L * D * 2, f"B{B} L{L} H{n_head} dh{dh}"):
            rc = lib.dfine_attention_bwd(_lib.ptr(qk), _lib.ptr(qk[..., D:]), _lib.ptr(v), D2, D2, D, _lib.ptr(mask), _lib.ptr(bits_q),
                                         _lib.ptr(bits_k), _lib.ptr(o), D, _lib.ptr(lse), _lib.ptr(go), D, _lib.ptr(dqk), D2, _lib.ptr(dqk[..., D:]), D2, _lib.ptr(dv), D,
                                         _lib.ptr(dsum), B, n_head, L, dh, scale, _lib.stream_ptr())
        _lib.check(rc, "dfine_attention_bwd")
        return dqk, dv, None, None


def self_attention_supported(qk, v, n_head):
    D = v.shape[-1]
    return (qk.is_cuda and qk.dtype == torch.bfloat16 and v.dtype == torch.bfloat16 and D % n_head == 0
            and (D // n_head) % 8 == 0 and D // n_head <= 64 and qk.shape[-1] == 2 * D)


def self_attention(qk, v, blocked, n_head):
    """qk [B, L, 2D] bf16 (query projection in columns [0, D), key projection in [D, 2D)), v [B, L, D] bf16, ``blocked``
    bool [L, L] (True = may not attend) or None -> softmax(q k^T / sqrt(dh) + mask) v as [B, L, D] bf16, heads interleaved
    the way nn.MultiheadAttention's out_proj expects. HIP forward and backward (csrc/attention.hip)."""
    return _SelfAttention.apply(qk, v, blocked, int(n_head))


class _ConvTranspose2x2(torch.autograd.Function):
    """nn.ConvTranspose2d(kernel 2, stride 2, no bias) on a channels_last map: forward = four 1-tap implicit-GEMM
    launches writing interleaved output pixels, input gradient = one 4-tap stride-2 gather, weight gradient = the
    split-M conv wgrad kernel with the roles of input and output gradient swapped."""

    @staticmethod
    def forward(ctx, x, weight, stats):
        from . import gemm

        xl = gemm.nhwc(x.to(torch.bfloat16))
        if gemm.pixel_pitch(xl) is None:
            xl = xl.contiguous()
        yl = gemm.conv_transpose2x2_forward(xl, bf16_t_of(weight), stats=stats)  # transposed copy: [CO, 2, 2, CI]
        ctx.save_for_backward(xl)
        ctx.weight = weight
        return yl.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, gy):
        from . import gemm

        (xl,) = ctx.saved_tensors
        weight = ctx.weight
        lib = _lib.load()
        CI, CO = weight.shape[0], weight.shape[1]
        gyl = gemm.nhwc(gy.to(torch.bfloat16))
        if gemm.pixel_pitch(gyl) is None:
            gyl = gyl.contiguous()
        B, H, W, _ = xl.shape
        dx = None
        if ctx.needs_input_grad[0]:
            dx = gemm.conv_transpose2x2_input_grad(gyl, _khwc(bf16_of(weight))).permute(0, 3, 1, 2)
        direct = (DIRECT_WGRAD_ACCUM and weight.grad is not None and weight.grad.dtype == torch.float32 and _is_khwc(weight.grad))
        dw = weight.grad if direct else torch.zeros(CI, CO, 2, 2, dtype=torch.float32, device=xl.device).contiguous(memory_format=torch.channels_last)
        M = B * H * W
        with _lib.timed("wgrad", (M * CI + gyl.numel()) * 2 + CI * CO * 16, f"convT {B}x{H}x{W} {CI}->{CO}"):
            rc = lib.dfine_conv_wgrad(_lib.ptr(xl), _lib.ptr(gyl), _lib.ptr(dw), B, 2 * H, 2 * W, CO, H, W, CI, 2, 2, 2, 0,
                                      gemm.pixel_pitch(xl), gemm.pixel_pitch(gyl), _lib.stream_ptr())
        _lib.check(rc, "dfine_conv_wgrad")
        return dx, (None if direct else dw.to(weight.dtype)), None


def conv_transpose2x2(x, weight, stats=None):
    """``F.conv_transpose2d(x, weight, None, stride=2)`` for a [CI, CO, 2, 2] weight on a channels_last map; bf16 GPU path
    on the HIP kernels (``stats``: see ``dense_conv``), otherwise the stock op."""
    bf16_mode = x.is_cuda and (torch.is_autocast_enabled() or x.dtype == torch.bfloat16)
    if not bf16_mode or tuple(weight.shape[2:]) != (2, 2):
        if stats is not None:
            raise _lib.DfineHipError("conv_transpose2x2: stats need the bf16 GPU path")
        return torch.nn.functional.conv_transpose2d(x, weight, None, 2)
    if not (torch.is_grad_enabled() and (weight.requires_grad or x.requires_grad)):
        from . import gemm

        return gemm.conv_transpose2x2_forward(gemm.nhwc(x.to(torch.bfloat16)), bf16_t_of(weight), stats=stats).permute(0, 3, 1, 2)
    return _ConvTranspose2x2.apply(x, weight, stats)


class _Embedding(torch.autograd.Function):
    """nn.Embedding lookup whose weight gradient is ONE HIP launch of float atomics into the fp32 gradient
    (dfine_embedding_wgrad) instead of ATen's sort / unique-by-key / segment pipeline."""

    @staticmethod
    def forward(ctx, idx, weight, padding_idx):
        ctx.save_for_backward(idx)
        ctx.weight, ctx.padding_idx = weight, padding_idx
        return weight.detach().index_select(0, idx.reshape(-1)).view(*idx.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        weight, padding_idx = ctx.weight, ctx.padding_idx
        lib = _lib.load()
        rows, D = weight.shape
        g = g.contiguous()
        if g.dtype not in (torch.float32, torch.bfloat16):
            g = g.float()
        direct = DIRECT_WGRAD_ACCUM and weight.grad is not None and weight.grad.dtype == torch.float32 and weight.grad.is_contiguous()
        dw = weight.grad if direct else torch.zeros(rows, D, dtype=torch.float32, device=g.device)
        flat = idx.reshape(-1).contiguous()
        with _lib.timed("embedding_wgrad", flat.numel() * D * (g.element_size() + 4)):
            rc = lib.dfine_embedding_wgrad(_lib.ptr(flat), _lib.ptr(g), _lib.dtype_code(g), _lib.ptr(dw), flat.numel(), D, rows,
                                           -1 if padding_idx is None else int(padding_idx), _lib.stream_ptr())
        _lib.check(rc, "dfine_embedding_wgrad")
        return None, (None if direct else dw.to(weight.dtype)), None


def embedding(idx, weight, padding_idx=None):
    """``F.embedding(idx, weight, padding_idx)`` (idx int64, any shape). On the GPU the backward is the HIP atomic
    kernel; on the CPU this is the stock op."""
    if not (idx.is_cuda and weight.is_cuda) or not (weight.requires_grad and torch.is_grad_enabled()):
        return torch.nn.functional.embedding(idx, weight, padding_idx)
    return _Embedding.apply(idx.to(torch.int64), weight, padding_idx)


def pointwise_conv(x, weight):
    """1x1 stride-1 ``dense_conv``."""
    return dense_conv(x, weight, 1, 0)
