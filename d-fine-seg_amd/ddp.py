"""Data-parallel gradient exchange for one process per GPU over RCCL (backend "nccl" on ROCm),
replacing the reference's ``DistributedDataParallel`` wrap (src/dl/train.py:157-169, collectives
C1-C3 in SURVEY.md 2.3).

  * gradients live in ONE flat fp32 buffer (``optim.FlatState``: ``p.grad`` are views into it, laid
    out in the order backward produces them); a bucket is a contiguous range of that buffer;
  * when the last gradient of a bucket has been accumulated, the range is all-reduced (mean) on a
    side HIP stream, overlapping the rest of backward; ``finish()`` joins before the optimizer;
  * xGMI is point-to-point (7 links per GPU): few large messages beat DDP's 25 MB default, so the
    default is 4 buckets for D-FINE-m's 78 MB of gradients;
  * parameters that received no gradient this step (e.g. the denoising embedding on a batch with no
    ground truth) contribute zeros -- the reference's ``find_unused_parameters=False`` DDP would
    raise on the following iteration instead;
  * parameters are broadcast from rank 0 once; BatchNorm statistics stay per-rank (DDP re-broadcasts
    all buffers every forward, C3).

Two exchange schedules over the same flat buffer:

  * ``FlatGradAllReduce`` (bench.py's default for N > 1): forward+matching, losses+backward and the
    optimizer tail are captured hipGraphs; between their replays run the criterion's 2-float all-reduce
    and ONE RCCL all-reduce that averages the whole gradient buffer (78 MB for D-FINE-m: ~2-3 ms over xGMI against a ~50 ms step). No overlap
    with backward, but no per-kernel launch cost either -- eager launching costs this model ~40 ms of
    host time per step, far more than the exposed collective;
  * ``BucketedGradReducer``: eager step, buckets reduced on a side stream as backward fills them.
"""
import os

import torch
import torch.distributed as dist

from .optim import FlatState


def init_distributed():
    """Process group from the torchrun environment (reference dist_utils.py:12-43): RCCL when a
    GPU is present, Gloo otherwise. Returns (rank, world_size, local_rank)."""
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        return 0, 1, 0
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", 0))
    if not dist.is_initialized():
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    return rank, world, local


def assign_buckets(entries, num_buckets=4, min_bucket_bytes=1 << 20):
    """Cut the flat layout (``FlatState.entries``: (name, param, offset, numel, group)) into at most
    ``num_buckets`` contiguous runs of roughly equal byte size. Returns lists of entries."""
    total = sum(e[3] * 4 for e in entries)
    target = max(total / max(num_buckets, 1), min_bucket_bytes)
    buckets, cur, cur_bytes = [], [], 0
    for e in entries:
        cur.append(e)
        cur_bytes += e[3] * 4
        if cur_bytes >= target and len(buckets) < num_buckets - 1:
            buckets.append(cur)
            cur, cur_bytes = [], 0
    if cur:
        buckets.append(cur)
    return buckets


class _Bucket:
    def __init__(self, entries, flat_grads):
        self.params = [e[1] for e in entries]
        begin = entries[0][2]
        end = entries[-1][2] + entries[-1][3]
        self.flat = flat_grads[begin:end]  # padding between optimizer groups is zero and stays zero
        self.pending = len(self.params)
        self.work = None


class BucketedGradReducer:
    """``reducer = BucketedGradReducer(model_or_flat_state)`` before training; per step: backward (hooks
    fire), ``reducer.finish()``, optimizer (which zeroes the flat gradients) or ``reducer.zero_grad()``."""

    def __init__(self, model, num_buckets=4, process_group=None, broadcast_parameters=True):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("BucketedGradReducer needs an initialised process group")
        from . import ops

        ops.DIRECT_WGRAD_ACCUM = False  # bucket completion is counted from autograd's accumulate hooks
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.flat = model if isinstance(model, FlatState) else FlatState(model)
        tensors = [self.flat.params]
        if not isinstance(model, FlatState):
            tensors += [b for b in model.buffers()] + [p for p in model.parameters() if not p.requires_grad]
        device = self.flat.device
        self.on_gpu = device.type == "cuda"
        self.side_stream = torch.cuda.Stream(device) if self.on_gpu else None
        self.backend_has_avg = dist.get_backend(process_group) == "nccl"
        if broadcast_parameters:  # C2: everyone starts from rank 0's weights (one flat message)
            with torch.no_grad():
                for t in tensors:
                    dist.broadcast(t, src=0, group=process_group)
            # the broadcast wrote the flat base in place, which does not move the parameter views' versions: the bf16
            # (and transposed) shadows built before it would silently keep this rank's pre-broadcast weights
            self.flat.refresh_bf16_shadow()
        self.buckets = [_Bucket(b, self.flat.grads) for b in assign_buckets(self.flat.entries, num_buckets)]
        self._owner = {}
        for b in self.buckets:
            for p in b.params:
                self._owner[p] = b
                p.register_post_accumulate_grad_hook(self._on_grad_ready)

    # ------------------------------------------------------------------ hooks
    def _on_grad_ready(self, param):
        bucket = self._owner[param]
        bucket.pending -= 1
        if bucket.pending == 0:
            self._launch(bucket)

    def _launch(self, bucket):
        if bucket.work is not None:
            return
        if self.on_gpu:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            self.side_stream.wait_event(ready)  # the bucket's last gradient has been written
            with torch.cuda.stream(self.side_stream):
                bucket.work = self._all_reduce(bucket.flat)
        else:
            bucket.work = self._all_reduce(bucket.flat)

    def _all_reduce(self, flat):
        if self.backend_has_avg:
            return dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        work.wait()
        flat.div_(self.world)
        return work

    # ------------------------------------------------------------------ per-step API
    def finish(self):
        """Reduce buckets whose gradients never all arrived (unused parameters count as zero), then
        make the current stream wait for every all-reduce."""
        for b in self.buckets:
            if b.work is None:
                self._launch(b)
        for b in self.buckets:
            b.work.wait()
            if self.on_gpu:
                torch.cuda.current_stream().wait_stream(self.side_stream)
            b.work, b.pending = None, len(b.params)

    def zero_grad(self):
        self.flat.grads.zero_()


class FlatGradAllReduce:
    """Average the flat gradient buffer over the ranks with one collective, between the captured
    forward/backward graph and the captured optimizer graph (``train.TrainStep`` replays
    graph A -> ``reduce()`` -> graph B). ``num_chunks`` > 1 splits the message (contiguous ranges)."""

    graph_safe = True  # TrainStep may capture around it: the collective itself is never inside a capture

    def __init__(self, model, num_chunks=1, process_group=None, broadcast_parameters=True):
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("FlatGradAllReduce needs an initialised process group")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.flat = model if isinstance(model, FlatState) else FlatState(model)
        tensors = [self.flat.params]
        if not isinstance(model, FlatState):
            tensors += [b for b in model.buffers()] + [p for p in model.parameters() if not p.requires_grad]
        self.backend_has_avg = dist.get_backend(process_group) == "nccl"
        if broadcast_parameters:
            with torch.no_grad():
                for t in tensors:
                    dist.broadcast(t, src=0, group=process_group)
            self.flat.refresh_bf16_shadow()
        n = self.flat.numel
        step = (n + num_chunks - 1) // max(num_chunks, 1)
        step = (step + 3) // 4 * 4
        self.chunks = [self.flat.grads[i : min(i + step, n)] for i in range(0, n, step)]

    def reduce(self):
        """Enqueue the all-reduce after the work already on the current stream; later work on that stream
        waits for it (stream order, the host does not block)."""
        for c in self.chunks:
            if self.backend_has_avg:
                dist.all_reduce(c, op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(c, op=dist.ReduceOp.SUM, group=self.group)
                c.div_(self.world)

    finish = reduce  # the name TrainStep's eager path calls

    def zero_grad(self):
        self.flat.grads.zero_()


def stage_runs(entries, prefixes):
    """Contiguous [begin, end) ranges of the flat layout per stage. Every entry must belong to exactly one stage (by name
    prefix); a run is a maximal sequence of NEIGHBOURS in offset order that share a stage, so the alignment padding between
    two of them rides along but no other stage's parameter ever lies inside a run (a foreign 1-element LAB scale between two
    same-stage tensors would otherwise be all-reduced on the side stream while its own stage is still writing it)."""
    def stage_of(name):
        hits = [i for i, p in enumerate(prefixes) if name.startswith(p)]
        if len(hits) != 1:
            raise RuntimeError(f"StagedGradAllReduce: parameter {name!r} belongs to {len(hits)} of the stages {prefixes}")
        return hits[0]

    runs = [[] for _ in prefixes]
    prev_stage = None
    for name, _, off, n, _ in sorted(entries, key=lambda e: e[2]):
        s = stage_of(name)
        if s == prev_stage:
            runs[s][-1][1] = off + n
        else:
            runs[s].append([off, off + n])
        prev_stage = s
    return [[tuple(r) for r in rs] for rs in runs]


class StagedGradAllReduce(FlatGradAllReduce):
    """The north-star schedule WITH graphs: backward is captured as three hipGraphs cut at the module boundaries
    (criterion + decoder | encoder | backbone); as soon as a stage's graph has been enqueued, the contiguous ranges of the
    flat gradient buffer that stage filled are all-reduced (mean) on a SIDE stream, while the next stage's graph replays
    on the main stream -- the collectives never sit inside a capture and only the last stage's message (the backbone's
    gradients, ~1/3 of the bytes) is exposed. ``TrainStep`` drives it (``staged = True``):

        graph(forward + matching) -> 2-float all-reduce -> graph(loss + decoder bwd) -> [side: all-reduce decoder ranges]
        -> graph(encoder bwd) -> [side: encoder ranges] -> graph(backbone bwd) -> [side: backbone ranges] -> join -> graph(optimizer)

    The flat layout makes this possible: inside each optimizer group the parameters are ordered by module in reverse
    registration order, so every (module, group) pair is ONE contiguous range (reference: DDP's reducer overlaps 25 MB
    buckets with an eagerly launched backward, src/dl/train.py:157-169)."""

    graph_safe = True
    staged = True
    stage_prefixes = ("decoder.", "encoder.", "backbone.")

    def __init__(self, model, process_group=None, broadcast_parameters=True, stage_prefixes=None):
        super().__init__(model, num_chunks=1, process_group=process_group, broadcast_parameters=broadcast_parameters)
        if stage_prefixes is not None:
            self.stage_prefixes = tuple(stage_prefixes)
        self.on_gpu = self.flat.device.type == "cuda"
        self.side_stream = torch.cuda.Stream(self.flat.device) if self.on_gpu else None
        self.stage_ranges = [[self.flat.grads[a:b] for a, b in runs]
                             for runs in stage_runs(self.flat.entries, self.stage_prefixes)]

    def reduce_stage(self, stage):
        """All-reduce the gradient ranges of ``stage`` on the side stream, after the work enqueued so far on the
        current stream (the stage's backward graph)."""
        if self.on_gpu:
            self.side_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side_stream):
                self._reduce_ranges(self.stage_ranges[stage])
        else:
            self._reduce_ranges(self.stage_ranges[stage])

    def _reduce_ranges(self, ranges):
        for r in ranges:
            if self.backend_has_avg:
                dist.all_reduce(r, op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(r, op=dist.ReduceOp.SUM, group=self.group)
                r.div_(self.world)

    def join(self):
        """The optimizer (current stream) waits for every stage's collective."""
        if self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.side_stream)

    def reduce(self):  # eager path (warm-up steps): everything at once, on the side stream, then join
        for stage in range(len(self.stage_ranges)):
            self.reduce_stage(stage)
        self.join()

    finish = reduce
