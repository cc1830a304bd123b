"""Flat parameter/gradient storage and the fused optimizer tail (HIP, csrc/optim.hip).

``FlatState`` re-homes every trainable parameter of a model into ONE fp32 buffer (and its
gradient into another), ordered by optimizer group; the parameters stay ordinary ``nn.Parameter``
views (same shapes, strides and ``state_dict``), so nothing else changes. On top of it:

  * ``FusedAdamWEMA.step()`` = global-norm clip + AdamW (the reference's four groups) + EMA +
    gradient zeroing in two kernel launches, with all per-step scalars read from a small device
    array (one async H2D copy per step) -- the launch sequence never changes, so the step can be
    captured in a hipGraph;
  * ``ddp.BucketedGradReducer`` uses contiguous ranges of the flat gradient buffer as its buckets.
"""
import math

import torch

from . import _lib

_HYPER_LEN = 24
_GROUP_OFF = 8


def group_parameters(model):
    """The reference's four AdamW groups (src/d_fine/dfine.py:87-124) as lists of (name, param)."""
    groups = {"backbone": [], "backbone_norm": [], "encdec_norm_bias": [], "rest": []}
    for name, p in model.named_parameters():
        is_norm = "norm" in name or "bn" in name
        if "backbone" in name:
            groups["backbone_norm" if is_norm else "backbone"].append((name, p))
        elif ("encoder" in name or "decoder" in name) and (is_norm or "bias" in name):
            groups["encdec_norm_bias"].append((name, p))
        else:
            groups["rest"].append((name, p))
    return groups


class FlatState:
    """Flat fp32 homes for the trainable parameters and their gradients.

    Order: group by group (``group_order``), and inside a group LAST registered first, so that a
    contiguous range of the buffer fills up in the order backward produces gradients."""

    def __init__(self, model, group_order=("rest", "encdec_norm_bias", "backbone", "backbone_norm")):
        groups = group_parameters(model)
        self.group_names = list(group_order)
        self.entries = []  # (name, param, offset, numel, group index)
        self.group_begin = []
        offset = 0
        for gi, gname in enumerate(group_order):
            self.group_begin.append(offset)
            for name, p in reversed(groups[gname]):
                if not p.requires_grad:
                    continue
                if p.dtype != torch.float32:
                    raise TypeError(f"{name}: fused optimizer expects fp32 master weights, got {p.dtype}")
                # every tensor starts on a multiple of 8 elements: its bf16 shadow (and transposed shadow) is then 16-byte
                # aligned, which the GEMM kernels' vector loads need (scalars such as the LAB scale/bias would otherwise
                # misalign everything behind them); the padding is zero in every flat buffer and stays zero
                offset = (offset + 7) // 8 * 8
                self.entries.append((name, p, offset, p.numel(), gi))
                offset += p.numel()
            offset = (offset + 7) // 8 * 8  # groups start aligned too
        self.numel = offset
        device = self.entries[0][1].device
        self.device = device
        self.params = torch.zeros(offset, dtype=torch.float32, device=device)
        self.grads = torch.zeros(offset, dtype=torch.float32, device=device)
        self.shadow = None  # bf16 copy of params, see enable_bf16_shadow()
        with torch.no_grad():
            for name, p, off, n, _ in self.entries:
                home = self.view_like(self.params, p, off)
                home.copy_(p.data)
                p.data = home
                p.grad = self.view_like(self.grads, p, off)

    def enable_bf16_shadow(self):
        """Keep a bf16 copy of every parameter in one flat buffer, refreshed by the fused optimizer
        kernel as it writes the fp32 value (2 more bytes per parameter in a launch that moves 40), and
        hang each parameter's view of it on the Parameter (``_bf16_shadow``): the bf16 compute path
        (``ops.linear`` / ``ops.dense_conv``) then reads the copy instead of casting the fp32 master
        weight in ~300 small launches per step. A parameter changed behind the optimizer's back (its
        ``_version`` moved: ``load_state_dict``, ``copy_``) is cast again until ``refresh_bf16_shadow``."""
        self.shadow = torch.empty(self.numel, dtype=torch.bfloat16, device=self.device)
        # transposed shadow: every GEMM weight (Linear [N, K]; dense conv / transposed conv in channels_last memory
        # [d0][kh][kw][d1]) also as [K][N] / [d1][kh][kw][d0] -- the operand of the input-gradient GEMMs -- rewritten by
        # ONE transpose launch after each optimizer step (dfine_transpose_weights)
        self.shadow_t = torch.empty(self.numel, dtype=torch.bfloat16, device=self.device)
        rows, blocks = [], []
        self._transposed = []
        for name, p, off, n, _ in self.entries:
            ktc = self.weight_ktc(p)
            if ktc is None:
                continue
            K, T, C = ktc
            tiles = T * ((K + 31) // 32) * ((C + 31) // 32)
            blocks.extend((len(rows), t) for t in range(tiles))
            rows.append((off, off, K, T, C))
            self._transposed.append((p, off, n, (C, K) if p.dim() == 2 else (C, p.shape[2], p.shape[3], K)))
        self._t_table = torch.tensor(rows, dtype=torch.int64, device=self.device) if rows else None
        self._t_blocks = torch.tensor(blocks, dtype=torch.int32, device=self.device) if rows else None
        self.refresh_bf16_shadow()

    @staticmethod
    def weight_ktc(p):
        """(K, T, C) when ``p``'s memory is a dense [K][T][C] GEMM weight: Linear [N, K_in] -> (N, 1, K_in); a
        channels_last conv weight [d0, d1, kh, kw] with d1 > 1 -> (d0, kh*kw, d1). None otherwise."""
        if p.dim() == 2 and p.is_contiguous():
            return p.shape[0], 1, p.shape[1]
        if p.dim() == 4 and p.shape[1] > 1:
            d0, d1, kh, kw = p.shape
            want = (kh * kw * d1, 1, kw * d1, d1)
            if all(p.shape[d] == 1 or p.stride(d) == want[d] for d in range(4)):
                return d0, kh * kw, d1
        return None

    def launch_transpose(self):
        """Rewrite the transposed shadow from the bf16 shadow (one launch, graph-capturable)."""
        if self.shadow is None or self._t_table is None:
            return
        lib = _lib.load()
        with _lib.timed("transpose_weights", self.numel * 4):
            rc = lib.dfine_transpose_weights(_lib.ptr(self.shadow), _lib.ptr(self.shadow_t), _lib.ptr(self._t_table),
                                             _lib.ptr(self._t_blocks), self._t_blocks.shape[0], _lib.stream_ptr())
        _lib.check(rc, "dfine_transpose_weights")

    def refresh_bf16_shadow(self):
        if self.shadow is None:
            return
        self.shadow.copy_(self.params)
        if self.device.type == "cuda":
            self.launch_transpose()
        for name, p, off, n, _ in self.entries:
            p._bf16_shadow = self.view_like(self.shadow, p, off)
            p._shadow_version = p._version
        if self.device.type == "cuda":
            for p, off, n, shape in self._transposed:
                p._bf16_shadow_t = self.shadow_t[off : off + n].view(shape)

    def bf16_shadow_is_current(self):
        return self.shadow is None or all(p._shadow_version == p._version for _, p, _, _, _ in self.entries)

    @staticmethod
    def view_like(flat, p, offset):
        """A view of ``flat`` with p's shape AND strides (channels_last conv weights stay channels_last)."""
        return flat[offset : offset + p.numel()].as_strided(p.size(), p.stride())

    def mirror(self, other_model):
        """Flat buffer holding ``other_model``'s copies of the same parameters, in the same order
        (used for the EMA model). Its parameters become views of the returned buffer."""
        theirs = dict(other_model.named_parameters())
        flat = torch.zeros(self.numel, dtype=torch.float32, device=self.device)
        with torch.no_grad():
            for name, p, off, n, _ in self.entries:
                q = theirs[name]
                home = self.view_like(flat, q, off)
                home.copy_(q.data)
                q.data = home
        return flat


class FusedAdamWEMA:
    """clip_grad_norm_ + AdamW + zero_grad (+ EMA of the weights) over a FlatState.

    ``lrs`` / ``weight_decays``: dicts by group name. ``set_schedule(lrs, ema_decay)`` before each
    ``step()`` refreshes the device-side scalars."""

    def __init__(self, flat: FlatState, lrs, weight_decays, betas=(0.9, 0.999), eps=1e-8, clip_max_norm=0.1,
                 ema_flat=None):
        self.flat, self.ema_flat = flat, ema_flat
        self.betas, self.eps, self.clip = betas, eps, clip_max_norm
        self.weight_decays = [float(weight_decays[g]) for g in flat.group_names]
        self.lrs = [float(lrs[g]) for g in flat.group_names]
        self.exp_avg = torch.zeros_like(flat.params)
        self.exp_avg_sq = torch.zeros_like(flat.params)
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=flat.device)
        self.hyper = torch.zeros(_HYPER_LEN, dtype=torch.float32, device=flat.device)
        # per-step scalars travel through a small RING of pinned buffers: with graph replay the host enqueues a step in
        # under a millisecond while the GPU takes tens, so a single staging buffer would be rewritten for step t+k before
        # step t's async copy has read it; a slot is reused only after the copy that read it has completed (event)
        on_gpu = flat.device.type == "cuda"
        self._hyper_ring = [torch.zeros(_HYPER_LEN, dtype=torch.float32).pin_memory() if on_gpu else torch.zeros(_HYPER_LEN)
                            for _ in range(8 if on_gpu else 1)]
        self._hyper_events = [None] * len(self._hyper_ring)
        self._group_begin = (_lib.ctypes.c_longlong * len(flat.group_begin))(*flat.group_begin)
        self.t = 0
        self.ema_decay = -1.0

    def set_schedule(self, lrs=None, ema_decay=None):
        if lrs is not None:
            self.lrs = [float(lrs[g]) for g in self.flat.group_names]
        if ema_decay is not None:
            self.ema_decay = float(ema_decay)

    def _upload_scalars(self):
        """Advance the step counter and push this step's scalars (one pinned async copy)."""
        self.t += 1
        slot = self.t % len(self._hyper_ring)
        if self._hyper_events[slot] is not None:
            self._hyper_events[slot].synchronize()  # the copy issued len(ring) steps ago
        h = self._hyper_ring[slot]
        h[0] = self.clip if self.clip else -1.0
        h[1], h[2], h[3] = self.betas[0], self.betas[1], self.eps
        h[4], h[5] = 1.0 - self.betas[0] ** self.t, 1.0 - self.betas[1] ** self.t
        h[6] = self.ema_decay if self.ema_flat is not None else -1.0
        for g, (lr, wd) in enumerate(zip(self.lrs, self.weight_decays)):
            h[_GROUP_OFF + 2 * g], h[_GROUP_OFF + 2 * g + 1] = lr, wd
        self.hyper.copy_(h, non_blocking=True)
        if self.flat.device.type == "cuda":
            ev = self._hyper_events[slot] or torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._hyper_events[slot] = ev

    def launch(self):
        """The device work of one step (graph-capturable: no host values are baked in)."""
        lib = _lib.load()
        f = self.flat
        st = _lib.stream_ptr()
        self.sumsq.zero_()
        with _lib.timed("grad_sumsq", f.numel * 4):
            _lib.check(lib.dfine_grad_sumsq(_lib.ptr(f.grads), f.numel, _lib.ptr(self.sumsq), st), "dfine_grad_sumsq")
        n_streams = 8 + (2 if self.ema_flat is not None else 0)  # p,g,m,v read and written (+ ema read and written)
        shadow_bytes = f.numel * 2 if f.shadow is not None else 0
        with _lib.timed("adamw_ema_step", f.numel * 4 * n_streams + shadow_bytes):
            rc = lib.dfine_adamw_ema_step(_lib.ptr(f.params), _lib.ptr(f.grads), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                                          _lib.ptr(self.ema_flat), _lib.ptr(f.shadow), _lib.ptr(self.hyper), _lib.ptr(self.sumsq),
                                          self._group_begin, len(f.group_begin), f.numel, st)
        _lib.check(rc, "dfine_adamw_ema_step")
        f.launch_transpose()  # the transposed bf16 shadow follows the bf16 shadow the kernel above rewrote

    def step(self):
        self._upload_scalars()
        self.launch()

    def grad_norm(self):
        """Total gradient norm of the LAST step (device tensor)."""
        return self.sumsq.sqrt().float()


def one_cycle_lr(step, total_steps, max_lr, pct_start=0.1, div_factor=25.0, final_div_factor=1e4):
    """torch.optim.lr_scheduler.OneCycleLR (cos annealing, two phases) as a pure function of the
    step index -- the reference's schedule (src/dl/train.py:193-208) without a stateful scheduler,
    so the learning rate can be fed to the fused optimizer as data."""
    initial, minimum = max_lr / div_factor, max_lr / div_factor / final_div_factor
    up_end = float(pct_start * total_steps) - 1
    last = total_steps - 1

    def cos(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1)

    step = min(step, last)
    if step <= up_end:
        return cos(initial, max_lr, step / up_end if up_end > 0 else 1.0)
    return cos(max_lr, minimum, (step - up_end) / (last - up_end))
