"""One optimizer step of the training loop, mirroring the reference's hot loop
(src/dl/train.py:488-562: autocast forward, criterion outside autocast, backward, clip 0.1,
AdamW over four groups, OneCycle LR, EMA) without its per-iteration host work (``loss.item()``,
an ``nvidia-smi`` subprocess, per-tensor target copies).

On the GPU the whole step -- forward, criterion with device-side matching, backward, fused
clip+AdamW+EMA -- issues no host synchronisation and reads every per-step scalar from device
memory, so after a few eager steps it is captured once into a hipGraph and replayed: the step is
otherwise bound by the ~9 000 kernel launches of the eager path, not by the GPU.
"""
import math
from copy import deepcopy

import os

import torch

from .dfine import build_loss, build_model
from .layers import ConvNorm
from .optim import FlatState, FusedAdamWEMA, one_cycle_lr
from .targets import TargetBatch


class ModelEMA:
    """Exponential moving average of the weights (train.py:51-72): decay
    momentum*(1-exp(-iters/2000)) over every floating-point state_dict entry. The parameters are
    averaged inside the fused optimizer kernel; this class owns the copy and averages the buffers
    (BatchNorm statistics etc.)."""

    def __init__(self, model, momentum=0.9998):
        self.model = deepcopy(model).eval()
        for p in self.model.parameters():
            p.requires_grad_(False)
        self.momentum = momentum
        # float buffers of both models are re-homed into two flat tensors (views keep their shapes), so
        # the buffer EMA is two launches instead of one per BatchNorm statistic
        self._ema_flat = self._flatten([b for _, b in self.model.named_buffers() if b.dtype.is_floating_point])
        self._cur_flat = self._flatten([b for _, b in model.named_buffers() if b.dtype.is_floating_point])

    @staticmethod
    def _flatten(tensors):
        if not tensors:
            return None
        flat = torch.empty(sum(t.numel() for t in tensors), dtype=torch.float32, device=tensors[0].device)
        off = 0
        with torch.no_grad():
            for t in tensors:
                home = flat[off : off + t.numel()].view(t.shape)
                home.copy_(t)
                t.data = home
                off += t.numel()
        return flat

    def decay(self, iters):
        return self.momentum * (1 - math.exp(-iters / 2000))

    def update_buffers(self, decay_tensor):
        """ema = decay*ema + (1-decay)*current for the float buffers; ``decay_tensor`` is a 0-dim device
        tensor so the same launches serve every step (mul/add form: anchors hold +inf, a lerp would NaN)."""
        if self._ema_flat is None:
            return
        with torch.no_grad():
            self._ema_flat.mul_(decay_tensor).addcmul_(self._cur_flat, 1.0 - decay_tensor)


class _SplitGraph:
    """Data-parallel step: graph(forward + matching) -> 2-float all-reduce -> graph(losses + backward) ->
    gradient all-reduce -> graph(optimizer), replayed as one step. The collectives stay outside the captures."""

    def __init__(self, head, reduce_norm, middle, reducer, tail):
        self.head, self.reduce_norm, self.middle, self.reducer, self.tail = head, reduce_norm, middle, reducer, tail

    def replay(self):
        self.head.replay()
        self.reduce_norm()
        self.middle.replay()
        self.reducer.reduce()
        self.tail.replay()


class _StagedGraph:
    """Data-parallel step with the gradient exchange overlapped: see ddp.StagedGradAllReduce."""

    def __init__(self, head, reduce_norm, stages, reducer, tail):
        self.head, self.reduce_norm, self.stages, self.reducer, self.tail = head, reduce_norm, stages, reducer, tail

    def replay(self):
        self.head.replay()
        self.reduce_norm()
        for i, g in enumerate(self.stages):
            g.replay()
            self.reducer.reduce_stage(i)  # side stream: overlaps the next stage's replay
        self.reducer.join()
        self.tail.replay()


PAD_IMAGE_CHANNELS = os.environ.get("DFINE_PAD_IMAGE", "1") != "0"  # bf16 path: feed the stem a 4-channel image


class TrainStep:
    """model + criterion + fused optimizer + OneCycle schedule + EMA; ``step(images, targets)`` runs
    one iteration and returns the (device-resident) total loss."""

    def __init__(self, model_name="m", num_classes=80, enable_mask_head=False, img_size=(640, 640),
                 device="cuda", base_lr=1.5e-4, backbone_lr=2e-5, betas=(0.9, 0.999), weight_decay=1.25e-4,
                 clip_max_norm=0.1, use_ema=True, ema_momentum=0.9998, amp=True, total_steps=10000,
                 pct_start=0.1, grad_reducer=None, use_graph=True, graph_warmup=3, graph_masks=True, overlap_wgrad=False, defer_wgrad=True,
                 sync_bn=False):
        self.device = torch.device(device)
        self.num_classes = num_classes
        self.model = build_model(model_name, num_classes, enable_mask_head, self.device, img_size=list(img_size))
        # reference train.py:159-160: under DDP with a per-GPU batch below 4 every BatchNorm2d becomes a SyncBatchNorm. The
        # fused norm kernels serve it (ops.batch_norm_act(sync=...): one all-reduce of the fp64 sums per norm and direction)
        self.sync_bn = bool(sync_bn)
        if self.sync_bn:
            self.model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(self.model)
        self.criterion = build_loss(model_name, num_classes, 0.0, enable_mask_head)
        self.ema = ModelEMA(self.model, ema_momentum) if use_ema else None
        self.flat = FlatState(self.model)
        if amp:
            self.flat.enable_bf16_shadow()
        ema_flat = self.flat.mirror(self.ema.model) if self.ema is not None else None
        # groups and max learning rates as in train.py:184-208: OneCycle peaks at 2x the configured lr -- per group
        # (backbone groups at 2*backbone_lr) only for the l / x sizes; for n / s / m the reference passes the SCALAR
        # max_lr = 2*base_lr, which OneCycleLR applies to all four groups, backbone included
        bb_max = backbone_lr * 2 if model_name in ("l", "x") else base_lr * 2
        self.max_lrs = {"backbone": bb_max, "backbone_norm": bb_max, "encdec_norm_bias": base_lr * 2, "rest": base_lr * 2}
        wds = {"backbone": weight_decay, "backbone_norm": 0.0, "encdec_norm_bias": 0.0, "rest": weight_decay}
        self.optimizer = FusedAdamWEMA(self.flat, self.max_lrs, wds, betas=betas, clip_max_norm=clip_max_norm, ema_flat=ema_flat)
        self.total_steps, self.pct_start = total_steps, pct_start
        self.amp = amp and self.device.type == "cuda"
        self.iters = 0
        self.grad_reducer = grad_reducer(self.flat) if grad_reducer is not None else None
        if self.grad_reducer is not None:  # non-trainable state is not in the flat buffer: align it once
            for t in list(self.model.buffers()) + [p for p in self.model.parameters() if not p.requires_grad]:
                torch.distributed.broadcast(t, src=0)
            # the EMA copy was taken from this rank's own initial weights BEFORE the reducer broadcast rank 0's: re-seed it
            # from the (now common) live weights and buffers so the averaged model is identical on every rank
            self.sync_ema_to_model()
        # (SyncBatchNorm issues its per-norm collectives from the host between kernels: such a step is not captured)
        self.use_graph = (use_graph and self.device.type == "cuda" and not self.sync_bn
                          and (self.grad_reducer is None or getattr(self.grad_reducer, "graph_safe", False)))
        self.graph_warmup = graph_warmup
        self.graph_masks = graph_masks
        self._graphs = {}  # (batch, max_gt, has_gt) -> (graph, static images, static TargetBatch, static loss)
        self._conv_norms = [m for m in self.model.modules() if isinstance(m, ConvNorm)]
        self.arena = None
        if self.device.type == "cuda":  # zero-initialised scratch for the fused ops: one memset per step
            from . import ops

            self.arena = ops.ZeroArena(8 << 20, self.device)  # owned by this step: its graphs hold pointers into it
        # optional: weight-gradient kernels on a side stream beside the rest of backward (ops.on_wgrad_stream). Measured on
        # D-FINE-m batch 32: 45.96 ms/step with the fork/join edges in the graph vs 44.6 without -- the two branches compete
        # for HBM and the graph's parallel-branch bookkeeping costs more than the overlapped tails give; off by default
        self.wgrad_stream = torch.cuda.Stream(self.device) if (self.device.type == "cuda" and overlap_wgrad) else None
        # weight gradients are queued during backward and launched grouped at its end (ops.WgradQueue): on by default
        self.wgrad_queue = None
        if self.device.type == "cuda" and defer_wgrad and not overlap_wgrad:
            from . import ops

            self.wgrad_queue = ops.WgradQueue()
        self._graph_pool = None   # every captured graph of this step shares one private memory pool
        self.max_graphs = 16      # bounded: each distinct (shape, GT count) key is one graph
        self.last_step_replayed = False  # whether the most recent step() replayed a captured graph
        self.model.train()

    def sync_ema_to_model(self):
        """EMA weights and float buffers := the live model's (used after a broadcast / checkpoint load)."""
        if self.ema is None:
            return
        with torch.no_grad():
            if self.optimizer.ema_flat is not None:
                self.optimizer.ema_flat.copy_(self.flat.params)
            live = dict(self.model.named_parameters())
            for n, p in self.ema.model.named_parameters():
                if not live[n].requires_grad:
                    p.copy_(live[n])
            live_b = dict(self.model.named_buffers())
            for n, b in self.ema.model.named_buffers():
                b.copy_(live_b[n])

    # ------------------------------------------------------------------ one step, device side
    def _forward_match(self, images, tb, inject=None, cuts=None):
        """Model forward and the matchings; returns (outputs, criterion state) -- no collective inside. ``cuts`` (a dict):
        run the three modules with the autograd graph CUT at their boundaries (detached leaves), recording what
        ``_staged_backward`` needs to chain the pieces -- same arithmetic, backward in three separately capturable parts."""
        from . import ops

        ops.zero_arena = self.arena  # this step's scratch (module-level handle: the ops are free functions)
        ops.wgrad_stream = self.wgrad_stream
        ops.wgrad_queue = self.wgrad_queue
        if self.wgrad_queue is not None:
            self.wgrad_queue.discard()  # leftovers of an abandoned backward must never reach this step's launch
        if self.arena is not None:
            self.arena.reset()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.amp):
            if cuts is None:
                out = self.model(images, targets=tb, inject=inject)
            else:
                m = self.model
                feats = m.backbone(images.contiguous(memory_format=torch.channels_last))
                feats_cut = [f.detach().requires_grad_(True) for f in feats]
                outs, inner = m.encoder(feats_cut)
                leaf = {}
                for t in list(outs) + list(inner):  # inner[0] IS outs[0]: one leaf per distinct tensor
                    if id(t) not in leaf:
                        leaf[id(t)] = (t, t.detach().requires_grad_(True))
                out = m.decoder([leaf[id(t)][1] for t in outs], [leaf[id(t)][1] for t in inner], tb, inject)
                cuts.update(feats=feats, feats_cut=feats_cut, enc=list(leaf.values()))
        return out, self.criterion.match(out)  # fp32, outside autocast like train.py:548-549

    @staticmethod
    def _backward_through(pairs):
        """Continue backward from cut leaves into the graph that produced them: pairs of (producer output, leaf)."""
        live = [(t, l.grad) for t, l in pairs if l.grad is not None and t.requires_grad]
        if live:
            torch.autograd.backward([t for t, _ in live], [g for _, g in live])
        from . import ops

        ops.join_wgrad()

    def _loss_backward(self, out, state):
        losses = self.criterion.compute_losses(out, state)
        self.last_losses = {k: v.detach() for k, v in losses.items()}  # the reference's loss dict of this step (device scalars)
        loss = torch.stack(list(losses.values())).sum()  # one reduction instead of a chain of ~50 adds
        loss.backward()
        from . import ops

        ops.join_wgrad()  # the weight gradients forked to the side stream during backward
        return loss.detach()

    def _forward_backward(self, images, tb, inject=None):
        out, state = self._forward_match(images, tb, inject)
        self.criterion.reduce_normalisers(state)  # the criterion's one collective (2 floats)
        return self._loss_backward(out, state)

    def _optimizer_tail(self):
        from . import ops

        ops.join_wgrad()  # every backward piece flushed already: nothing may be left in the queue
        ops.wgrad_queue = None  # the step's backward is over: ops used outside this step launch their own weight gradients
        self.optimizer.launch()  # clip + AdamW + EMA(params) + bf16 shadow + zero grads
        if self.ema is not None:
            self.ema.update_buffers(self.optimizer.hyper[6])
        # ... and so is this step's claim on the scratch arena: an op called outside a TrainStep (or by another TrainStep's
        # replayed graph, which dirties the arena without moving ``used``) must get its own zeroed workspace
        ops.zero_arena = None

    def _body(self, images, tb, inject=None):
        loss = self._forward_backward(images, tb, inject)
        if self.grad_reducer is not None:
            self.grad_reducer.finish()
        self._optimizer_tail()
        return loss

    def _schedule(self):
        lrs = {g: one_cycle_lr(self.iters, self.total_steps, m, self.pct_start) for g, m in self.max_lrs.items()}
        self.iters += 1
        decay = self.ema.decay(self.iters) if self.ema is not None else None
        self.optimizer.set_schedule(lrs, decay)
        self.optimizer._upload_scalars()

    @staticmethod
    def _inject_leaves(inject):
        """(key path, tensor) pairs of an ``inject`` dict ({"topk_ind": t, "cdn_noise": {name: t}}) in a fixed order."""
        leaves = []
        for k in sorted(inject or {}):
            v = inject[k]
            if isinstance(v, dict):
                leaves += [((k, kk), v[kk]) for kk in sorted(v)]
            elif v is not None:
                leaves.append(((k,), v))
        return leaves

    def step(self, images, targets, inject=None):
        """``inject`` (parity runs): {"topk_ind": int64 [B, Q], "cdn_noise": {...}} replaces the query selection and the
        denoising noise of this step (see DFINETransformer.forward); under graph replay its tensors are copied into static
        buffers like the images and targets."""
        tb = TargetBatch.from_list(targets, self.num_classes, self.device)
        images = images.to(self.device, non_blocking=True)
        self._schedule()
        if not self.flat.bf16_shadow_is_current():  # weights were written behind the optimizer's back (load_state_dict, ...)
            self.flat.refresh_bf16_shadow()
        if tb.masks is not None:  # per-image resize of the GT masks to the mask-logit size: host loop, outside any capture
            s = self.model.decoder.spec.strides[0] // 2  # mask logits live at half the finest stride (ConvTranspose 2x2)
            tb.dense_mask_targets(images.shape[-2] // s, images.shape[-1] // s, self.device)
        # segment task: the step with the mask branch is dense and sync-free like the detect step and is captured too
        # (``graph_masks=False`` keeps mask batches eager). Round 1's captured D-FINE-m segment runs faulted inside ATen's
        # embedding_dense_backward (its sort / unique-by-key scratch pipeline); the denoising embedding's
        # backward is now one HIP kernel (ops.embedding) and that pipeline is no longer on the path
        if not self.use_graph or self.iters <= self.graph_warmup or (tb.mask_tgt is not None and not self.graph_masks):
            self.last_step_replayed = False
            return self._body(self._model_input(images), tb, inject)
        leaves = self._inject_leaves(inject)
        key = (tuple(images.shape), tb.max_gt, max(tb.num_gts), tb.total > 0, tb.mask_tgt is not None,
               tuple((path, tuple(t.shape)) for path, t in leaves))
        entry = self._graphs.get(key)
        if entry is None:
            if len(self._graphs) >= self.max_graphs:  # drop the least recently used graph
                self._graphs.pop(next(iter(self._graphs)))
            entry = self._capture(images, tb, inject)
        else:
            self._graphs.pop(key)
        self._graphs[key] = entry  # most recently used last
        self.last_step_replayed = True
        graph, s_images, s_tb, s_loss, s_leaves = entry
        for (_, dst), (_, src) in zip(s_leaves, leaves):
            dst.copy_(src, non_blocking=True)
        (s_images[:, :3] if s_images.shape[1] != images.shape[1] else s_images).copy_(images, non_blocking=True)
        s_tb.labels.copy_(tb.labels, non_blocking=True)
        s_tb.boxes.copy_(tb.boxes, non_blocking=True)
        s_tb.valid.copy_(tb.valid, non_blocking=True)
        s_tb.counts.copy_(tb.counts, non_blocking=True)
        if s_tb.mask_tgt is not None:
            s_tb.mask_tgt.copy_(tb.mask_tgt, non_blocking=True)
        graph.replay()
        from . import ops

        ops.zero_arena = None  # the replay wrote this step's arena; nothing outside the graph may take "pre-zeroed" slices of it
        ops.wgrad_queue = None
        for m in self._conv_norms:  # host-side batch counters the replay does not run
            m._pending_batches += 1
        return s_loss

    def _model_input(self, images):
        """The image batch as the stem's implicit GEMM reads it: bf16 under autocast (the conv casts its input anyway; doing it
        here makes the replay path's copy into the static buffer the one and only cast kernel)."""
        if self.amp and images.is_cuda and images.shape[1] == 3 and PAD_IMAGE_CHANNELS:
            # ... and zero-padded to 4 channels (8-byte pixels): the stem's 3-channel gather is element-wise otherwise
            B, _, H, W = images.shape
            out = torch.zeros(B, H, W, 4, dtype=torch.bfloat16, device=images.device).permute(0, 3, 1, 2)
            out[:, :3].copy_(images)
            return out
        if self.amp and images.dtype == torch.float32:
            return images.to(torch.bfloat16)
        return images

    def _capture(self, images, tb, inject=None):
        """See ``_capture_impl``. Python's cyclic GC is held off for the duration: a hipGraph object of an earlier TrainStep
        (or any other tensor-owning cycle) finalised in the middle of a capture calls hipGraphDestroy / hipFree while the
        stream is capturing -- "operation not permitted when stream is capturing", raised from a destructor = abort."""
        import gc

        gc.collect()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            if self.wgrad_queue is not None:
                self.wgrad_queue.reserve(self.device)  # persistent memory for the launch tables: not the graph's pool
            captured = self._capture_impl(images, tb, inject)
            if self.wgrad_queue is not None:
                self.wgrad_queue.upload_pending()  # the launch tables the captured grouped wgrad launches read on replay
            return captured
        finally:
            if was_enabled:
                gc.enable()

    def _capture_impl(self, images, tb, inject=None):
        """Capture the step for this (batch shape, padded GT width, max GT count). The per-image GT
        counts inside may vary between replays; the maximum (it fixes the number of denoising
        queries) may not -- a new maximum gets its own graph."""
        s_images = self._model_input(images).clone()  # bf16 under autocast: the per-replay copy into it is the ONLY cast
        s_tb = TargetBatch(tb.labels.clone(), tb.boxes.clone(), tb.valid.clone(), list(tb.num_gts), tb.counts.clone(), None,
                           tb.mask_tgt.clone() if tb.mask_tgt is not None else None)
        s_leaves = [(path, t.to(self.device).clone()) for path, t in self._inject_leaves(inject)]
        s_inject = None
        if s_leaves:
            s_inject = {}
            for path, t in s_leaves:
                if len(path) == 1:
                    s_inject[path[0]] = t
                else:
                    s_inject.setdefault(path[0], {})[path[1]] = t
        graph = torch.cuda.CUDAGraph()
        pool_kw = {"pool": self._graph_pool} if self._graph_pool is not None else {}
        # With a process group alive, c10d's watchdog THREAD polls hipEventQuery on in-flight collectives; under the default
        # "global" capture mode HIP forbids that call process-wide while any thread captures ("operation not permitted when
        # stream is capturing"), the watchdog's exception is uncaught and the process aborts. Captures are therefore
        # thread-local in data-parallel runs (kernels the autograd thread launches on the capturing stream are still captured).
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            pool_kw["capture_error_mode"] = "thread_local"
            torch.cuda.synchronize()  # and no collective of the eager warm-up steps is still in flight
        if self.grad_reducer is None:
            with torch.cuda.graph(graph, **pool_kw):
                s_loss = self._body(s_images, s_tb, s_inject)
            if self._graph_pool is None:
                self._graph_pool = graph.pool()
            return graph, s_images, s_tb, s_loss, s_leaves
        # data parallel: several graphs; the collectives (2-float normaliser all-reduce of the criterion, gradient
        # all-reduces) run between their replays and are never part of a capture
        staged = getattr(self.grad_reducer, "staged", False)
        cuts = {} if staged else None
        with torch.cuda.graph(graph, **pool_kw):
            out, state = self._forward_match(s_images, s_tb, s_inject, cuts)
        if self._graph_pool is None:
            self._graph_pool = graph.pool()
        reduce_norm = lambda: self.criterion.reduce_normalisers(state)
        if staged:  # backward in three graphs cut at the module boundaries; each stage's gradients leave on the side stream
            stages = [torch.cuda.CUDAGraph() for _ in range(3)]
            with torch.cuda.graph(stages[0], **{**pool_kw, 'pool': graph.pool()}):
                s_loss = self._loss_backward(out, state)  # criterion + decoder; fills the cut leaves' .grad
            with torch.cuda.graph(stages[1], **{**pool_kw, 'pool': graph.pool()}):
                self._backward_through(cuts["enc"])  # encoder
            with torch.cuda.graph(stages[2], **{**pool_kw, 'pool': graph.pool()}):
                self._backward_through(list(zip(cuts["feats"], cuts["feats_cut"])))  # backbone
            tail = torch.cuda.CUDAGraph()
            with torch.cuda.graph(tail, **{**pool_kw, 'pool': graph.pool()}):
                self._optimizer_tail()
            return _StagedGraph(graph, reduce_norm, stages, self.grad_reducer, tail), s_images, s_tb, s_loss, s_leaves
        middle = torch.cuda.CUDAGraph()
        with torch.cuda.graph(middle, **{**pool_kw, 'pool': graph.pool()}):
            s_loss = self._loss_backward(out, state)
        tail = torch.cuda.CUDAGraph()
        with torch.cuda.graph(tail, **{**pool_kw, 'pool': graph.pool()}):
            self._optimizer_tail()
        return _SplitGraph(graph, reduce_norm, middle, self.grad_reducer, tail), s_images, s_tb, s_loss, s_leaves
