#!/usr/bin/env python3
"""Headline benchmark: OSVOS-VGG one-shot online fine-tune throughput in 854x480 frames/s.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one iteration of the reference's online loop body (src/train_online.py:70-101) on one
synthetic 1x3x480x854 frame already resident in HBM: forward (5 logit maps) -> class-balanced BCE on
the fused map -> /avg_grad_every_n -> backward -> fused SGD step + zero_grad every 5th iteration.  It runs
through the drop-in ``train_online._train`` with the drop-in ``OSVOS_VGG`` module, i.e. the shipped path.

N > 1 (one process per GPU, RCCL): data-parallel fine-tuning - every rank runs the same K steps on its
own frames (weak scaling, per-GPU work fixed) and the flat fp32 gradient buffer (59.7 MB) is
SUM-all-reduced over xGMI before each optimizer step.  ``--mode replicas`` instead runs N independent
sequences with no collective (the reference's -sg/-sgs sharding).

Prints ONE JSON line on rank 0 with the contract fields plus
  roofline     : MFMA roofline of the conv3x3 implicit-GEMM kernels (fwd + dgrad + wgrad), per-launch
                 durations measured with HIP events on the launch stream in a separate pass after
                 the timed region; `by_kernel` lists every kernel family of the iteration
  cpu_baseline : the CPU oracle (a port of the reference's arithmetic in plain torch fp32) timed on this
                 box's host cores on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))

H, W = 480, 854
AVG_GRAD_EVERY_N = 5
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
CONV_TRAIN_GFLOP_PER_FRAME = 773.27  # fwd + dgrad + wgrad of the 3x3 convs, no dgrad into the image


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", choices=["dp", "replicas"], default="dp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=3)
    return ap.parse_args()


def cpu_baseline(iters: int):
    """The oracle's online loop on the host cores: 1 warm-up + `iters` timed fwd+bwd iterations of the same
    480x854 workload (SGD step every 5th iteration included when it falls inside the sample)."""
    import torch
    from oracle import osvos_ref as O
    # the GPU box gives one GPU's job a 16-core share of the host; more threads than that only oversubscribe
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(share, 16)))
    sd = O.make_state_dict(9)
    x, gt = O.synthetic_frame(1, H, W, seed=1234)
    params = O.leaf_params(sd)
    opt = O.make_sgd(params, "online")

    def one(i):
        outs = O.forward(params, x)
        loss = O.cbce_loss(outs[-1], gt, size_average=False)
        (loss / AVG_GRAD_EVERY_N).backward()
        if (i + 1) % AVG_GRAD_EVERY_N == 0:
            opt.step()
            opt.zero_grad()

    one(-1)  # warm-up (oneDNN primitive creation)
    t0 = time.perf_counter()
    for i in range(iters):
        one(i)
    dt = time.perf_counter() - t0
    return {"value": iters / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 warm-up + {iters} timed fwd+loss+bwd iterations at 1x3x{H}x{W} fp32 (oracle/osvos_ref.py, "
                      f"torch {torch.__version__} CPU), {dt / iters * 1000:.0f} ms/iter"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import fosvos_amd  # noqa: F401
    import parallel
    import train_online
    from dataloaders.synthetic import make_frame
    from fosvos_hip import ops
    from networks.osvos_vgg import OSVOS_VGG
    from util.network_provider import VGGOnlineProvider

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus > 1:
            raise SystemExit(f"--gpus {args.gpus} needs a torch.distributed.run launch with {args.gpus} ranks "
                             f"(WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one process per GPU; FOSVOS_DIST_BACKEND=gloo lets several ranks share one GPU for rehearsals on a 1-GPU box
    backend = os.environ.get("FOSVOS_DIST_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            parallel.init_distributed("nccl")
        else:
            dist.init_process_group(backend=backend)

    # ---- model: seeded random-init weights of the real architecture (no checkpoints offline)
    torch.manual_seed(1234)
    net = OSVOS_VGG(pretrained=0)
    # variance-preserving init so activations/gradients are O(1)-O(100) like a trained net (the reference's
    # N(0,1e-3) init collapses every activation to ~0 after 13 layers)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.startswith("upscale"):
                continue
            if p.dim() == 4:
                fan_in = p.shape[1] * p.shape[2] * p.shape[3]
                p.normal_(0, (2.0 / fan_in) ** 0.5 if name.startswith("stages") else (1.0 / fan_in) ** 0.5)
            else:
                p.normal_(0, 0.1)
    net = net.to(dev)
    prov = VGGOnlineProvider.__new__(VGGOnlineProvider)
    prov.network = net
    prov.name = "vgg16"
    opt = prov.get_optimizer()
    img, gt = make_frame(H, W, seed=1234, index=rank)
    batch = [{"image": img.unsqueeze(0).to(dev), "gt": gt.unsqueeze(0).to(dev)}]  # resident in HBM
    train_online.data_parallel = world > 1 and args.mode == "dp"
    accum = AVG_GRAD_EVERY_N * (world if train_online.data_parallel else 1)  # each rank accumulates 5 micro-batches

    def run(n_steps):
        return train_online._train(prov, batch, opt, _NullWriter(), "bench", 0, n_steps, accum, 10 ** 9)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    frames = args.steps * world
    out = {
        "metric": "online fine-tune frames/sec (854x480)",
        "value": frames / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1000.0,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": "OSVOS_VGG train_online one-shot fine-tune: 1x3x480x854 frame per step, fwd + class-balanced "
                        "BCE + bwd, fused SGD step every 5 steps (BASELINE.json configs[1])",
            "frame": [H, W], "batch": 1, "avg_grad_every_n": AVG_GRAD_EVERY_N,
            "activations": "bf16 NHWC, fp32 accumulate, fp32 master weights",
            "parallelism": (f"dp{world}: flat fp32 gradient all-reduce (59.7 MB, RCCL) per optimizer step"
                            if train_online.data_parallel else
                            (f"{world} independent replicas, no collective" if world > 1 else "single GPU")),
        },
    }

    if rank == 0 and not args.no_roofline:
        # rank-0-only pass: it must not issue collectives (the other ranks are already at the final barrier)
        dp_was = train_online.data_parallel
        train_online.data_parallel = False
        from fosvos_hip import engine
        native_was = engine.USE_NATIVE_LOOP
        engine.USE_NATIVE_LOOP = False  # same kernels, issued op by op from Python so that each op can be bracketed
        prof = ops.OpProfiler()
        ops.set_profiler(prof)
        n_prof = 5
        train_online._train(prov, batch, opt, _NullWriter(), "bench", 0, n_prof, AVG_GRAD_EVERY_N, 10 ** 9)
        ops.set_profiler(None)
        engine.USE_NATIVE_LOOP = native_was
        train_online.data_parallel = dp_was
        agg = prof.summary()
        by_kernel = {}
        for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
            per_iter_ms = a["ms"] / n_prof
            by_kernel[name] = {
                "launches_per_step": a["calls"] / n_prof,
                "ms_per_step": round(per_iter_ms, 4),
                "avg_us_per_launch": round(a["ms"] / a["calls"] * 1000.0, 2),
                "tflops": round(a["flops"] / a["ms"] / 1e9, 2) if a["flops"] else None,
                "gbs": round(a["bytes"] / a["ms"] / 1e6, 1),
            }
        conv = [agg[k] for k in ("conv3x3_fwd", "conv3x3_dgrad", "conv3x3_wgrad") if k in agg]
        conv_ms = sum(a["ms"] for a in conv) / n_prof
        conv_calls = sum(a["calls"] for a in conv) / n_prof
        conv_flop = sum(a["flops"] for a in conv) / n_prof
        achieved = conv_flop / (conv_ms * 1e-3) / 1e12
        # HBM bytes per launch of the same kernel family from the committed PMC passes (rocprofv3 cannot run inside
        # this process); null when no summary is present
        traffic, traffic_note = None, "no profiles/*_pmc_traffic.json present"
        try:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
            if cands:
                pm = json.load(open(cands[-1]))
                traffic = pm["summary"]["conv_mfma_family"]["hbm_bytes_per_launch"]
                traffic_note = (f"bytes/launch, conv MFMA family, from {os.path.relpath(cands[-1], ROOT)}: {pm['source']}; "
                                f"{pm['corrections']}")
        except Exception as e:  # a malformed summary must not break the bench line
            traffic_note = f"could not read PMC summary: {e}"
        out["roofline"] = {
            "bound": "mfma",
            "kernel": "conv3x3 implicit-GEMM family (k_conv3x3_igemm fwd+dgrad, k_wgrad), bf16 MFMA 16x16x32",
            "achieved": achieved,
            "peak": MFMA_BF16_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / MFMA_BF16_PEAK_TFLOPS,
            "traffic": traffic,
            "traffic_note": traffic_note,
            "launches_per_step": conv_calls,
            "algorithmic_gflop_per_step": conv_flop / 1e9,
            "avg_launch_us": conv_ms / conv_calls * 1000.0,
            "device_ms_per_step_all_kernels": sum(a["ms"] for a in agg.values()) / n_prof,
            "by_kernel": by_kernel,
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_iters)
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass

    def close(self):
        pass


if __name__ == "__main__":
    main()
