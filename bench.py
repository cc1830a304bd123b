"""Benchmark of the D-FINE training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N=1)
    python bench.py --gpus N ...                              (N>1: starts N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = forward (bf16 autocast) + criterion with device-side Hungarian matching + backward
(+ gradient all-reduce over RCCL when N>1) + clip + AdamW + OneCycle + EMA on one synthetic
COCO-shaped batch already resident in HBM. Rank 0 prints ONE JSON line: the BASELINE.json metric
(images/sec, D-FINE-m 640x640 bf16), the roofline of the dominant hand-written kernel measured
with HIP events inside the timed region, and a CPU baseline (the fp32 oracle on the host cores).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak, same guide (not the 2:1-sparsity headline figure)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="m")
    ap.add_argument("--img", type=int, default=640)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (weak scaling)")
    ap.add_argument("--gts", type=int, default=8, help="ground truths per image (8 => 192 denoising queries)")
    ap.add_argument("--task", default="detect", choices=["detect", "segment"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="run every step eagerly (no hipGraph capture)")
    ap.add_argument("--no-miopen-find", action="store_true", help="(no effect any more: no library convolution is on the path)")
    ap.add_argument("--bucketed-reducer", action="store_true",
                    help="N>1: eager step + bucketed all-reduce from autograd hooks (default: captured graphs, backward in three stages "
                         "whose gradient ranges are all-reduced on a side stream while the next stage replays)")
    ap.add_argument("--flat-reducer", action="store_true",
                    help="N>1: captured graphs with ONE flat all-reduce between backward and optimizer (no overlap)")
    ap.add_argument("--force-ddp", action="store_true",
                    help="diagnostic: take the N>1 code path (process group, reducer, split graphs) even with one rank")
    ap.add_argument("--no-graph-masks", action="store_true", help="segment task: run mask batches eagerly (default: captured like detect)")
    ap.add_argument("--shapes", action="store_true", help="diagnostic: key the per-kernel timings by operand shape")
    return ap.parse_args()


def cpu_baseline(model, args, num_classes=80):
    """The fp32 CPU oracle (oracle/, a port of the reference's PyTorch path) on the host cores:
    full step (forward + criterion + backward + clip + AdamW) on a bounded sample of the workload."""
    from oracle import configs as ocfg
    from oracle import criterion as ocrit
    from oracle import model as omodel

    from dfine_seg_amd.synth import synthetic_batch

    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    trainable = {n for n, p in model.named_parameters() if p.requires_grad}
    params = []
    for k in sd:
        if k in trainable:
            sd[k].requires_grad_(True)
            params.append(sd[k])
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1.25e-4)
    cfg = ocfg.size_cfg(args.model)
    B = 2
    seg = args.task == "segment"
    images, targets = synthetic_batch(B, args.img, args.img, num_classes, args.gts, with_masks=seg, seed=7)

    def one_step():
        out = omodel.forward(sd, cfg, images, targets, training=True, enable_mask_head=seg)
        loss = sum(ocrit.criterion(out, targets, num_classes, with_masks=seg).values())
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 0.1)
        opt.step()
        opt.zero_grad(set_to_none=True)

    one_step()  # warm-up
    n, t0 = 0, time.perf_counter()
    while n < 2 or (time.perf_counter() - t0 < 10 and n < 8):
        one_step()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": B * n / dt, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{n} full fp32 steps of D-FINE-{args.model} {args.img}x{args.img} at batch {B} "
                      f"({args.gts} GT/img) through oracle/ on the host CPU"}


TIMER_TO_DEVICE_KERNEL = {
    "bn_stats": "bn_stats_kernel", "bn_apply": "bn_apply_kernel", "bn_bwd_reduce": "bn_bwd_reduce_kernel",
    "bn_bwd_dx": "bn_bwd_dx_kernel", "wgrad": "wgrad_group_kernel", "msda_fwd": "msda_fwd_kernel", "msda_bwd": "msda_bwd_kernel",
    "dwconv_fwd": "dwconv_strip_kernel", "dwconv_bwd_data": "dwconv_strip_kernel", "dwconv_bwd_weight": "dwconv_bwd_weight_kernel",
    "criterion_head": "criterion_head_kernel", "grad_sumsq": "grad_sumsq_kernel", "adamw_ema_step": "adamw_ema_kernel",
    "fdr_head_fwd": "fdr_head_fwd_kernel", "fdr_head_bwd": "fdr_head_bwd_kernel", "stem_pool_fwd": "pool_fwd_kernel",
    "stem_pool_bwd": "pool_bwd_kernel", "add_layernorm_fwd": "add_ln_fwd_kernel", "add_layernorm_bwd": "add_ln_bwd_kernel",
    "mask_loss_stats": "mask_stats_kernel", "mask_loss_grad": "mask_grad_kernel",
    # the implicit GEMM's two bodies: gathers (k x k / strided convs) and unaligned shapes on the register-staged igemm_kernel,
    # plain rows (1x1 convs, Linears, the mask contraction) on the LDS-DMA ring igemm2_kernel
    "conv_fwd": "igemm_kernel", "conv_dgrad": "igemm_kernel", "gemm_fwd": "igemm_kernel", "gemm_dgrad": "igemm_kernel",
    "conv_fwd_rows": "igemm2_kernel", "conv_dgrad_rows": "igemm2_kernel", "gemm_fwd_rows": "igemm2_kernel",
    "gemm_dgrad_rows": "igemm2_kernel", "mask_rows_stats": "igemm2_kernel", "mask_rows_dz": "igemm2_kernel",
    "attention_fwd": "attn_fwd_kernel", "attention_bwd": "attn_bwd_dkv_kernel", "transpose_weights": "transpose_weights_kernel",
    "embedding_wgrad": "embedding_wgrad_kernel", "msda_bwd_points": "msda_bwd_kernel", "msda_bwd_value": "msda_bwd_value_kernel",
}


def load_pmc_traffic():
    """Newest profiles/*pmc_traffic.json (written by tools/pmc_traffic.sh on the GPU box and committed)."""
    import glob

    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*pmc_traffic.json")))
    if not files:
        return {}, None
    with open(files[-1]) as f:
        return json.load(f), "profiles/" + os.path.basename(files[-1])


def spawn_ranks(n_ranks):
    """`python bench.py --gpus N` with no torchrun environment: start N rank processes of this same script (one per GPU,
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as torch.distributed.run would, rendezvous on 127.0.0.1) BEFORE this
    process has touched the GPU, relay rank 0's stdout (its single JSON line) and exit with the ranks' status. The reference
    starts its ranks with `torchrun --nproc_per_node=N` (Makefile:21-23); this is that launcher for the driver's plain
    command shape. DFINE_BENCH_WORKER names another worker script (tests use a stub)."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    worker = os.environ.get("DFINE_BENCH_WORKER", os.path.abspath(__file__))
    if "DFINE_BENCH_WORKER" not in os.environ:  # compile once here (hipcc only, no GPU call) instead of N ranks queueing on the build lock
        import __graft_entry__

        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)  # whatever a compile prints goes to stderr: stdout carries rank 0's JSON line and nothing else
        try:
            __graft_entry__.build()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
        # rank 0 inherits this process's stdout; the other ranks' stdout goes to stderr so exactly one line reaches stdout
        procs.append(subprocess.Popen([sys.executable, worker] + sys.argv[1:], env=env, stdout=None if r == 0 else sys.stderr))
    rc, kill_at = 0, None
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:  # a dead rank leaves the others waiting in a collective: end them (exact PIDs)
                    q.terminate()
                kill_at = time.monotonic() + 20
        if kill_at is not None and time.monotonic() > kill_at:
            for q in alive:
                q.kill()
            kill_at = None
        time.sleep(0.05)
    return rc if rc > 0 else (1 if rc else 0)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))
    # stdout carries exactly ONE line (the JSON): RCCL prints a version banner to stdout when the first communicator is
    # created, so everything else this process writes to fd 1 is sent to stderr and the JSON goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_ddp:  # one MIOpen user db / kernel cache per rank: N ranks tuning at once would contend on one sqlite file
        os.environ.setdefault("MIOPEN_USER_DB_PATH", f"/tmp/miopen_udb_rank{rank}")
        os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", f"/tmp/miopen_cache_rank{rank}")
        for var in ("MIOPEN_USER_DB_PATH", "MIOPEN_CUSTOM_CACHE_DIR"):
            os.makedirs(os.environ[var], exist_ok=True)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or args.force_ddp:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=device)

    import __graft_entry__

    __graft_entry__.build()
    from dfine_seg_amd import _lib
    from dfine_seg_amd.synth import synthetic_batch
    from dfine_seg_amd.train import TrainStep

    seg = args.task == "segment"
    reducer = None
    if world > 1 or args.force_ddp:
        from dfine_seg_amd import ddp

        # default: hipGraphs with backward cut into decoder / encoder / backbone stages, each stage's gradient ranges
        # all-reduced over RCCL on a side stream while the next stage's graph replays (ddp.StagedGradAllReduce);
        # --flat-reducer: one flat all-reduce after backward; --bucketed-reducer: eager step with autograd-hook buckets
        reducer = (ddp.BucketedGradReducer if args.bucketed_reducer else
                   ddp.FlatGradAllReduce if args.flat_reducer else ddp.StagedGradAllReduce)
    torch.manual_seed(42 + rank)  # per-rank seed like train.py:131
    trainer = TrainStep(args.model, 80, seg, (args.img, args.img), device, total_steps=max(1000, args.steps + args.warmup + 1),
                        grad_reducer=reducer, use_graph=not args.no_graph, graph_masks=not args.no_graph_masks,
                        sync_bn=reducer is not None and args.batch < 4)  # reference train.py:159-160
    images, targets = synthetic_batch(args.batch, args.img, args.img, 80, args.gts, with_masks=seg, seed=42 + rank, device=device)
    images = images.contiguous(memory_format=torch.channels_last)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, trainer.graph_warmup + 1 if trainer.use_graph else 0)):
        trainer.step(images, targets)  # eager steps (MIOpen tuning, caches), then the hipGraph capture
    sync()
    t0 = time.perf_counter()
    replayed = 0
    for _ in range(args.steps):
        loss = trainer.step(images, targets)
        replayed += int(trainer.last_step_replayed)
    sync()
    elapsed = time.perf_counter() - t0
    # Per-kernel durations: HIP events around every hand-written launch, on the launch stream. The timed
    # steps above replay a captured graph (no Python between kernels to record events from), so the same
    # step is run eagerly a few more times right after the timed region, on the same data and weights.
    probe_steps = 3
    timer = _lib.KernelTimer(by_detail=args.shapes)
    graph_was_on, trainer.use_graph = trainer.use_graph, False
    _lib.set_kernel_timer(timer)
    from dfine_seg_amd import ops as _ops

    _ops.bnred_counts.update(fused=0, discarded=0)
    for _ in range(probe_steps):
        trainer.step(images, targets)
    sync()
    _lib.set_kernel_timer(None)
    if rank == 0:  # norm backwards whose reduction pass a GEMM epilogue replaced / epilogue sums that went unused, per step
        print("bn backward reductions per step:", {k: v // probe_steps for k, v in _ops.bnred_counts.items()}, file=sys.stderr)
    trainer.use_graph = graph_was_on
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss_value = float(loss)
    if not (loss_value == loss_value):
        raise SystemExit("non-finite loss in the benchmark run")

    if rank == 0:
        spec = trainer.model.decoder.spec
        Lq = spec.num_queries + 2 * args.gts * max(1, spec.num_denoising // max(args.gts, 1))
        ksum = timer.summary()
        kernels = {}
        for name, st in ksum.items():
            gbs = st["avg_bytes"] / (st["avg_us"] * 1e-6) / 1e9
            kernels[name] = {"launches": st["launches"], "avg_us": round(st["avg_us"], 2),
                             "algorithmic_bytes": int(st["avg_bytes"]), "achieved_GBs": round(gbs, 1),
                             "ms_per_step": round(st["total_ms"] / probe_steps, 3)}
            if st["total_flops"]:
                kernels[name]["achieved_TFLOPs"] = round(st["total_flops"] / (st["total_ms"] * 1e-3) / 1e12, 1)
        # HBM-side bytes per launch from the committed PMC passes of this same workload (tools/pmc_traffic.sh:
        # separate FETCH_SIZE and WRITE_SIZE runs, FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950)
        pmc, pmc_file = load_pmc_traffic()
        for name, k in kernels.items():
            dev = TIMER_TO_DEVICE_KERNEL.get(name.split(" ")[0])
            k["traffic_bytes"] = pmc[dev]["traffic_bytes"] if dev in pmc else None
        # the hand-written DEVICE kernel the step spends most time in (timer names that share a kernel are added up: the
        # implicit-GEMM kernel serves conv forward / input gradient and Linear forward / input gradient). Its roofline:
        # MFMA when its algorithmic FLOP/s are the larger fraction of their peak, HBM otherwise; `achieved` = algorithmic
        # work of all its launches / their summed HIP-event durations (= per-launch work / average launch duration).
        fam = {}
        for name, st in ksum.items():
            dev = TIMER_TO_DEVICE_KERNEL.get(name.split(" ")[0], name.split(" ")[0])
            f = fam.setdefault(dev, {"launches": 0, "total_ms": 0.0, "bytes": 0.0, "flops": 0.0, "call_sites": []})
            f["launches"] += st["launches"]
            f["total_ms"] += st["total_ms"]
            f["bytes"] += st["avg_bytes"] * st["launches"]
            f["flops"] += st["total_flops"]
            f["call_sites"].append(name)
        dominant = max(fam, key=lambda n: fam[n]["total_ms"])
        d = fam[dominant]
        secs = d["total_ms"] * 1e-3
        gbs, tfs = d["bytes"] / secs / 1e9, d["flops"] / secs / 1e12
        mfma_bound = tfs / MFMA_PEAK_TFLOPS > gbs / HBM_PEAK_GBS
        roofline = {"kernel": dominant, "call_sites": sorted(set(c.split(" ")[0] for c in d["call_sites"])),
                    "bound": "mfma" if mfma_bound else "hbm",
                    "achieved": round(tfs if mfma_bound else gbs, 1), "peak": MFMA_PEAK_TFLOPS if mfma_bound else HBM_PEAK_GBS,
                    "unit": "TFLOP/s" if mfma_bound else "GB/s",
                    "frac": round((tfs / MFMA_PEAK_TFLOPS) if mfma_bound else (gbs / HBM_PEAK_GBS), 4),
                    "launches_per_step": d["launches"] // probe_steps, "avg_launch_us": round(1e3 * d["total_ms"] / d["launches"], 2),
                    "ms_per_step": round(d["total_ms"] / probe_steps, 3),
                    "algorithmic_GBs": round(gbs, 1), "algorithmic_TFLOPs": round(tfs, 1),
                    "traffic": pmc[dominant]["traffic_bytes"] if dominant in pmc else None, "traffic_source": pmc_file,
                    "timing": f"HIP events over {probe_steps} eager steps run right after the timed region "
                              "(the timed steps replay a hipGraph)" if graph_was_on else "HIP events over eager steps",
                    "kernels": kernels}
        result = {
            "metric": f"images/sec training, D-FINE-{args.model} {args.img}x{args.img} bf16",  # default = BASELINE.json's metric
            "value": round(args.batch * world * args.steps / elapsed, 2),
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"D-FINE-{args.model} {args.task} {args.img}x{args.img}, per-GPU batch {args.batch}, "
                                   f"{args.gts} GT/img ({Lq} decoder queries), COCO-80 synthetic",
                       "global_batch": args.batch * world, "parallelism": f"dp{world}"},
            "loss": round(loss_value, 4),
            "hip_graph": replayed == args.steps,  # every timed step replayed a captured graph
            "graph_replays": replayed,
            "peak_device_memory_GB": round(torch.cuda.max_memory_allocated(device) / 2**30, 2),
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(trainer.model, args)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
