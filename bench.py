#!/usr/bin/env python3
"""Headline benchmark: OSVOS-VGG one-shot online fine-tune throughput in 854x480 frames/s.

  python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: under ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` (the ranks
come from the environment) and as a plain ``python bench.py --gpus N``, which then starts N fresh child processes of
itself, one per GPU, BEFORE it touches the GPU or imports torch (it never re-executes a process that did).

One "step" = one iteration of the reference's online loop body (src/train_online.py:70-101) on one synthetic
1x3x480x854 frame already resident in HBM: forward -> class-balanced BCE on the fused map -> /avg_grad_every_n ->
backward -> fused SGD step + zero_grad every 5th iteration.  It runs through the drop-in ``train_online._train`` with
the drop-in ``OSVOS_VGG`` module, i.e. the shipped path (native layer loop, two-stream backward, deferred wgrad join).

N > 1, one process per GPU over RCCL.  Three mappings of BASELINE.json configs[3] are timed in the same run:
  dp        (``value``) every rank runs the same K steps on its own frames and the flat fp32 gradient buffer is
            SUM-all-reduced (bucketed, overlapped with the backward pass) before each optimizer step: weak scaling,
            5 local micro-batches per rank and step
  dp_strict (``dp_strict`` object) the parity-preserving split of ONE sequence's fine-tune (SURVEY.md section 8(e)(ii),
            src/train_online.py:92-101): world x local_accum = avg_grad_every_n, with avg_grad_every_n the smallest
            multiple of the world size >= 5 (8 ranks: 8, one micro-batch per rank and optimizer step) - the mode in which
            the 59.7 MB all-reduce has to hide behind ONE frame per rank
  replicas  (``replicas`` object) N independent sequences, no collective: the reference's own -sg/-sgs sharding
            (src/train_online.py:178-189), the parity-preserving mapping
The line carries ``backend``, ``ranks`` and ``distinct_devices``; ``n_gpus`` is the number of DISTINCT devices, and
the nccl backend is refused when two ranks share one (a gloo rehearsal of two ranks on one GPU reads n_gpus 1).

Extra objects on the one JSON line rank 0 prints (N = 1):
  cold            the same K steps behind the W warm-up steps only (no preconditioning): what a short window measures on a
                  device that is still raising its clocks; timed FIRST, right after the model is built
  group1          one frame per pass (FOSVOS_MICROBATCH_GROUP=1: the reference's one-by-one order, and what a cycle of five
                  different frame sizes degenerates to)
  mixed_scales    the reference's real augmentation: frames drawn from {1.0, 0.8, 0.5} x 480x854 with a fixed seed
                  (src/dataloaders/custom_transforms.py:63-76), bucketed by shape inside each accumulation cycle
  offline         BASELINE.json configs[2]: train_offline._train on batches of 16 x 480x854 frames, five deeply supervised
                  losses, avg_grad_every_n = 10 (src/train_offline.py:77-110)
  roofline        MFMA roofline of the conv3x3 kernels (k_conv3x3_igemm forward + data gradient, k_wgrad3x3 weight
                  gradient): per-launch durations from HIP events recorded by the library around every kernel ON ITS
                  OWN LAUNCH STREAM (fosvos_profile_start/stop), over `--prof-steps` further steps of the same shipped
                  execution right behind the timed region; `by_kernel` lists every kernel of the step under the name
                  rocprofv3 prints (profiles/ holds the --kernel-trace --stats summary of this same command)
  cpu_baseline    the CPU oracle (a port of the reference's arithmetic in plain torch fp32) on this box's host cores:
                  2 warm-up + 10 timed iterations of the same workload, rank 0, N = 1 only
  infer           the reference's own timing protocol for its only published number (src/util/experiment_helper.py:
                  29-53,77-80: net.forward between device syncs, 10 passes, first frame of each pass dropped)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))

H, W = 480, 854
AVG_GRAD_EVERY_N = 5
# untimed steps of the same loop in front of the W warm-up steps: ~60 ms of load, so that the timed window does not start on a
# device that is still raising its clocks after the idle start-up of the process (FOSVOS_BENCH_PRECONDITION=0 turns it off)
PRECONDITION_STEPS = int(os.environ.get("FOSVOS_BENCH_PRECONDITION", "70"))
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
# the MFMA kernels the roofline object is about (forward: persistent k_conv3x3_pp or k_conv3x3_igemm; data gradient:
# k_conv3x3_igemm; weight gradient: k_wgrad3x3_v2, k_wgrad_first)
CONV_KERNEL_PREFIXES = ("k_conv3x3_igemm", "k_conv3x3_pp", "k_wgrad3x3", "k_wgrad_first")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", choices=["both", "dp", "dp_strict", "replicas"], default="both",
                    help="N > 1: which mapping(s) to time (both = dp, dp_strict and replicas); `value` is dp unless only "
                         "another one is asked for")
    ap.add_argument("--no-variants", action="store_true", help="skip the cold / group1 / mixed_scales / offline objects")
    ap.add_argument("--no-alone", action="store_true",
                    help="skip roofline.alone (the conv kernels one at a time): its launches would sit in a rocprofv3 summary "
                         "of this command beside the in-step ones")
    ap.add_argument("--offline-batch", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-infer", action="store_true")
    ap.add_argument("--prof-steps", type=int, default=20)
    ap.add_argument("--cpu-iters", type=int, default=10)
    return ap.parse_args()


def launch_children(args) -> int:
    """`python bench.py --gpus N` outside torchrun: N fresh children, one per GPU.  This parent has not imported torch
    and makes no GPU call; it only waits and forwards exit codes."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FOSVOS_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # wait for all; a rank that fails takes the others down with it (they would wait at the rendezvous or in a collective)
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            rc = max(rc, abs(code))
            if code != 0:
                for q in live:
                    q.terminate()
    return rc


def cpu_baseline(iters: int):
    """The oracle's online loop on the host cores: 2 warm-up + `iters` timed fwd+bwd iterations of the same 480x854
    workload, the SGD step of every 5th iteration inside the sample."""
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

    for i in range(2):
        one(-2 + i)  # warm-up (oneDNN primitive creation); no optimizer step falls in here
    opt.zero_grad()
    t0 = time.perf_counter()
    for i in range(iters):
        one(i)
    dt = time.perf_counter() - t0
    return {"value": iters / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"2 warm-up + {iters} timed fwd+loss+bwd iterations (SGD step every {AVG_GRAD_EVERY_N}th) at "
                      f"1x3x{H}x{W} fp32 (oracle/osvos_ref.py, torch {torch.__version__} CPU), {dt / iters * 1000:.0f} ms/iter"}


def alone_rates(dev, n_frames, H, W):
    """The conv MFMA kernels ALONE on the chip: every 3x3 layer shape of the backbone (conv1_2 .. conv5_3) at `n_frames`
    frames per launch, one op at a time, HIP events on the launch stream; FLOP-weighted TFLOP/s of the forward, data-gradient
    and weight-gradient launches.  What the kernels sustain when nothing shares the chip with them - the in-step durations
    of `roofline.by_kernel` are those of kernels that overlap (two forward chains, data- and weight-gradient streams)."""
    import torch
    from fosvos_hip import ops
    shapes = [(1, 64, 64, 1), (2, 64, 128, 1), (2, 128, 128, 1), (4, 128, 256, 1), (4, 256, 256, 2), (8, 256, 512, 1),
              (8, 512, 512, 2), (16, 512, 512, 3)]  # (downscale, Ci, Co, layers of that shape)
    g = torch.Generator(device=dev).manual_seed(7)
    tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}

    def timeit(fn, reps=5):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    for ds, ci, co, count in shapes:
        h, w = H, W
        d = 1
        while d < ds:  # ceil-mode pooling
            h, w, d = (h + 1) // 2, (w + 1) // 2, d * 2
        x = torch.randn(n_frames, h, w, ci, device=dev, generator=g).to(torch.bfloat16)
        dy = torch.randn(n_frames, h, w, co, device=dev, generator=g).to(torch.bfloat16)
        wt = torch.randn(co, ci, 3, 3, device=dev, generator=g) * (2.0 / (9 * ci)) ** 0.5
        b = torch.zeros(co, device=dev)
        wf, wd = ops.pack_conv3x3_weights(wt)
        flop = 2.0 * n_frames * h * w * 9 * ci * co * count
        for op, fn in (("fwd", lambda: ops.conv3x3_fwd(x, wf, b, ci, co, relu=True)),
                       ("dgrad", lambda: ops.conv3x3_dgrad(dy, wd, ci, co, relu_src=x)),
                       ("wgrad", lambda: ops.conv3x3_wgrad(x, dy, ci, co))):
            tot[op][0] += flop
            tot[op][1] += timeit(fn) * count
        del x, dy
    res = {op: round(f / t / 1e12, 1) for op, (f, t) in tot.items()}
    all_f, all_t = sum(v[0] for v in tot.values()), sum(v[1] for v in tot.values())
    return {"frames_per_launch": n_frames, "tflops": res, "family_tflops": round(all_f / all_t / 1e12, 1),
            "family_frac": all_f / all_t / 1e12 / MFMA_BF16_PEAK_TFLOPS,
            "note": "12 backbone layers (conv1_2..conv5_3), one op at a time, weight gradient incl. its slab reduction"}


def offline_config(args, dev, make_frame, barrier):
    """BASELINE.json configs[2]: train_offline._train itself on resident batches of 16 synthetic 480x854 frames, five deeply
    supervised losses, avg_grad_every_n = 10 (src/train_offline.py:77-110).  One epoch = 10 iterations = one optimizer
    step; 1 warm-up epoch, 2 timed epochs."""
    import torch
    import train_offline
    from networks.osvos_vgg import OSVOS_VGG
    from util.network_provider import VGGOfflineProvider
    nb = args.offline_batch
    torch.manual_seed(4321)
    net = OSVOS_VGG(pretrained=0)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.startswith("upscale"):
                continue
            if p.dim() == 4:
                fan_in = p.shape[1] * p.shape[2] * p.shape[3]
                p.normal_(0, (2.0 / fan_in) ** 0.5 if name.startswith("stages") else (1.0 / fan_in) ** 0.5)
            else:
                p.normal_(0, 0.1)
    prov = VGGOfflineProvider.__new__(VGGOfflineProvider)
    prov.network = net.to(dev)
    prov.name = "vgg16"
    opt = prov.get_optimizer()
    frames = [make_frame(H, W, seed=4321, index=i) for i in range(nb)]
    batch = {"image": torch.stack([f[0] for f in frames]).to(dev), "gt": torch.stack([f[1] for f in frames]).to(dev)}
    accum, timed_epochs = 10, 2
    loader = [batch] * accum
    train_offline.data_parallel = False

    def run(first_epoch, n):
        return train_offline._train(prov, loader, None, opt, _NullWriter(), first_epoch, first_epoch + n, accum, 10 ** 9, False, 5)

    run(0, 1)
    barrier()
    t0 = time.perf_counter()
    ret = run(1, timed_epochs)
    barrier()
    e = time.perf_counter() - t0
    iters = ret["iterations"]
    conv_flop = 773.27e9 * nb  # fwd + dgrad + wgrad of the 3x3 convs per 480x854 frame
    return {"value": iters * nb / e, "unit": "frames/s", "ms_per_step": e / iters * 1000.0, "iterations": iters,
            "batch": nb, "avg_grad_every_n": accum, "losses": 5,
            "step_wall_tflops": conv_flop * iters / e / 1e12,
            "step_wall_frac": conv_flop * iters / e / 1e12 / MFMA_BF16_PEAK_TFLOPS,
            "workload": f"train_offline._train: {nb}x3x{H}x{W} per iteration, five class-balanced BCE losses "
                        f"((1 - epoch/n_epochs) * side losses + fused), backward, fused SGD step every {accum} iterations "
                        f"(BASELINE.json configs[2]); one iteration is one step"}


def _with_steps(args, steps, warmup, fn):
    saved = args.steps, args.warmup
    args.steps, args.warmup = steps, warmup
    try:
        return fn()
    finally:
        args.steps, args.warmup = saved


_SAMPLER = r"""
import os, re, subprocess, sys, time
exe, path = sys.argv[1], sys.argv[2]
with open(path, "w") as f:
    try:
        txt = subprocess.run([exe, "static", "--limit"], capture_output=True, text=True, timeout=10).stdout
        m = re.search(r"SOCKET_POWER_LIMIT:\s*(\d+)\s*W", txt)
        f.write("limit %s\n" % (m.group(1) if m else "-"))
    except Exception:
        f.write("limit -\n")
    f.flush()
    while True:
        if not os.path.exists(path + ".go"):  # asleep outside the power window: the other timed regions see no sampler
            time.sleep(0.05)
            continue
        try:
            txt = subprocess.run([exe, "metric", "--power", "--clock"], capture_output=True, text=True, timeout=10).stdout
            w = re.search(r"SOCKET_POWER:\s*(\d+)\s*W", txt)
            clk = [int(m) for m in re.findall(r"GFX_\d+:\s*\n\s*CLK:\s*(\d+)\s*MHz", txt)]
            if w and clk:
                f.write("%.3f %s %.1f %d\n" % (time.time(), w.group(1), sum(clk) / len(clk), min(clk)))
                f.flush()
        except Exception:
            pass
        time.sleep(0.25)
"""


def _start_power_sampler():
    """A child process that - while the file <path>.go exists - asks amd-smi (read-only) for socket power and GFX clocks
    four times a second and appends them to a file.  Started BEFORE this process touches the GPU (a process that has initialised the GPU starts no programs);
    None when amd-smi is not there."""
    import shutil
    import tempfile
    exe = shutil.which("amd-smi") or ("/opt/rocm/bin/amd-smi" if os.path.exists("/opt/rocm/bin/amd-smi") else None)
    if exe is None:
        return None
    fd, path = tempfile.mkstemp(prefix="fosvos_power_", suffix=".txt")
    os.close(fd)
    try:
        proc = subprocess.Popen([sys.executable, "-c", _SAMPLER, exe, path], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except OSError:
        return None
    return proc, path


def _stop_power_sampler(sampler):
    if sampler is None:
        return
    proc, path = sampler
    proc.terminate()
    try:
        proc.wait(timeout=15)
    except subprocess.TimeoutExpired:
        proc.kill()
    for f in (path, path + ".go"):
        try:
            os.unlink(f)
        except OSError:
            pass


def _power_summary(sampler, t_begin, t_end, n_steps, elapsed):
    """The sampler's lines between the wall-clock times t_begin and t_end."""
    if sampler is None:
        return None
    limit, busy = None, []
    try:
        with open(sampler[1]) as f:
            for line in f:
                parts = line.split()
                if parts and parts[0] == "limit":
                    limit = int(parts[1]) if parts[1].isdigit() else None
                elif len(parts) == 4 and t_begin <= float(parts[0]) <= t_end:
                    busy.append((int(parts[1]), float(parts[2]), int(parts[3])))
    except (OSError, ValueError):
        return None
    if len(busy) < 2:
        return None
    return {"socket_w_avg": sum(r[0] for r in busy) / len(busy), "socket_w_max": max(r[0] for r in busy),
            "socket_w_limit": limit, "gfx_mhz_avg": sum(r[1] for r in busy) / len(busy),
            "gfx_mhz_min_xcd": min(r[2] for r in busy), "samples": len(busy), "steps": n_steps,
            "value": n_steps / elapsed, "unit": "frames/s",
            "note": "amd-smi metric --power --clock (a child process started before the GPU was touched) while the shipped "
                    "step runs for `steps` steps - its own timed window, not the headline's; clocks averaged over the XCDs"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        sys.exit(launch_children(args))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}, or run "
                         f"`python bench.py --gpus {args.gpus}` outside torchrun and let it start its own ranks")

    only_early = os.environ.get("FOSVOS_BENCH_ONLY")
    power_sampler = None
    if world == 1 and not args.no_variants and (only_early is None or "power" in only_early.split(",")):
        power_sampler = _start_power_sampler()
        import atexit
        atexit.register(_stop_power_sampler, power_sampler)

    import torch
    import torch.distributed as dist
    import fosvos_amd  # noqa: F401
    import parallel
    import train_online
    from dataloaders.synthetic import make_frame
    from fosvos_hip import LaunchProfile
    from networks.osvos_vgg import OSVOS_VGG
    from util.network_provider import VGGOnlineProvider

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # one process per GPU over RCCL; FOSVOS_DIST_BACKEND=gloo lets several ranks share a GPU for rehearsals on a 1-GPU box
    backend = os.environ.get("FOSVOS_DIST_BACKEND", "nccl") if world > 1 else "none"
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= n_dev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} GPU(s) visible - RCCL needs one device per "
                         f"rank (FOSVOS_DIST_BACKEND=gloo rehearses several ranks on one GPU)")
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # the library's two auxiliary streams, created before anything else in this process creates streams (the process group,
    # RCCL): which hardware queue a stream gets depends on creation order, and two streams on one queue do not overlap
    from fosvos_hip import engine as _engine
    _engine.shared_stream(dev_index, "comm")  # (creates all three, in order)
    distinct = 1
    if world > 1:
        dist.init_process_group(backend=backend)
        props = torch.cuda.get_device_properties(dev_index)
        ident = (socket.gethostname(), str(getattr(props, "uuid", dev_index)), dev_index)
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        distinct = len(set(idents))
        if backend == "nccl" and distinct != world:
            raise SystemExit(f"{world} RCCL ranks on {distinct} distinct device(s): refusing to run")

    # ---- model: seeded random-init weights of the real architecture (no checkpoints offline)
    def make_provider():
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
        prov = VGGOnlineProvider.__new__(VGGOnlineProvider)
        prov.network = net.to(dev)
        prov.name = "vgg16"
        return prov, prov.get_optimizer()

    img, gt = make_frame(H, W, seed=1234, index=rank)
    batch = [{"image": img.unsqueeze(0).to(dev), "gt": gt.unsqueeze(0).to(dev)}]  # resident in HBM

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def strict_avg():
        """avg_grad_every_n of the strict split: the smallest multiple of the world size that is >= the reference's 5."""
        return world * -(-AVG_GRAD_EVERY_N // world)

    def timed(mode, precondition=PRECONDITION_STEPS, loader=None, env=None):
        """W warm-up + exactly K timed steps of `mode`, barrier + device sync on both sides, MAX over ranks."""
        prov, opt = make_provider()
        train_online.data_parallel = world > 1 and mode in ("dp", "dp_strict")
        if mode == "dp_strict":
            accum = strict_avg()  # world x local_accum = avg_grad_every_n: the single-process update, split over the ranks
        else:
            accum = AVG_GRAD_EVERY_N * (world if train_online.data_parallel else 1)  # 5 local micro-batches per rank
        frames = loader if loader is not None else batch
        saved_env = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})

        def run(n_steps):
            if n_steps <= 0:
                return {}
            n_epochs, rest = divmod(n_steps, len(frames))
            assert rest == 0, (n_steps, len(frames))
            return train_online._train(prov, frames, opt, _NullWriter(), "bench", 0, n_epochs, accum, 10 ** 9)

        # Device wake-up, in front of the W warm-up steps (untimed, declared in the JSON line): after an idle gap - process
        # start-up, model construction - the device's clocks take a few milliseconds of load to come up, which a short timed
        # window would otherwise measure (tools/train_call_probe.py: 20 steps take 19.2 ms after 300 ms of idle, 17.4 ms
        # right behind another call).  The same workload, `precondition` steps of it.
        try:
            run(precondition)
            run(args.warmup)
            barrier()
            t0 = time.perf_counter()
            ret = run(args.steps)
            barrier()
            elapsed = time.perf_counter() - t0
        finally:
            for k, v in saved_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        host_lead[mode] = ret.get("seconds_host_enqueue")
        comm_timing[mode] = ret.get("comm_timing")
        if world > 1:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        train_online.data_parallel = False
        return elapsed, prov, opt, run

    comm_timing = {}  # per data-parallel mode: parallel.GradSync.timing_summary() of the timed steps
    parallel.COMM_TIMING = world > 1
    host_lead = {}  # per mode: seconds the host loop needed to ENQUEUE the timed steps (the device finishes later)
    modes = ["single"] if world == 1 else (["dp", "dp_strict", "replicas"] if args.mode == "both" else [args.mode])
    results = {}
    cold = None
    only = os.environ.get("FOSVOS_BENCH_ONLY")  # lab: comma-separated subset of cold,group1,mixed_scales,offline
    want = (lambda name: only is None or name in only.split(","))
    if world == 1 and not args.no_variants and want("cold"):
        # FIRST, on the device as the start-up of the process left it: W warm-up steps, K timed steps, no preconditioning
        cold = timed("cold", precondition=0)[0]
    for m in modes:
        results[m] = timed(m)
    head = modes[0]
    elapsed, prov, opt, run = results[head]

    def parallelism(m):
        if m == "single":
            return "single GPU"
        if m == "dp":
            return (f"dp{world}: 5 local micro-batches per rank and step, bucketed fp32 gradient all-reduce (59.7 MB per "
                    f"optimizer step, {backend}) overlapped with the backward pass")
        if m == "dp_strict":
            return (f"dp{world} strict: avg_grad_every_n = {strict_avg()} split over {world} ranks ({strict_avg() // world} "
                    f"micro-batch(es) per rank and optimizer step): the update of one process running all of them "
                    f"(SURVEY.md section 8(e)(ii)); bucketed all-reduce over {backend} behind every pass that closes a cycle")
        return f"{world} independent replicas (one sequence per rank, src/train_online.py:178-189), no collective"

    out = {
        "metric": "online fine-tune frames/sec (854x480)",
        "value": args.steps * world / elapsed,
        "unit": "frames/s",
        "n_gpus": distinct,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1000.0,
        "host_enqueue_ms_per_step": (host_lead.get(head) or 0.0) / args.steps * 1000.0,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "precondition_steps": PRECONDITION_STEPS,
        "backend": {"nccl": "nccl (RCCL)"}.get(backend, backend),
        "ranks": world,
        "distinct_devices": distinct,
        "config": {
            "workload": "OSVOS_VGG train_online one-shot fine-tune: 1x3x480x854 frame per step, fwd + class-balanced "
                        "BCE + bwd, fused SGD step every 5 steps (BASELINE.json configs[1])",
            "frame": [H, W], "batch": 1, "avg_grad_every_n": AVG_GRAD_EVERY_N,
            "activations": "bf16 NHWC, fp32 accumulate, fp32 master weights",
            "parallelism": parallelism(head),
        },
    }
    # every FOSVOS_* variable of this process (lab switches of the Python side; the shipped library itself reads none)
    out["env_overrides"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith("FOSVOS_")}
    if comm_timing.get(head):
        out["comm"] = comm_timing[head]
        out["comm_exposed_ms_per_step"] = comm_timing[head]["comm_exposed_ms_per_step"]
    for m in modes[1:]:
        e = results[m][0]
        out[m] = {"value": args.steps * world / e, "unit": "frames/s", "ms_per_step": e / args.steps * 1000.0,
                  "parallelism": parallelism(m)}
        if comm_timing.get(m):
            out[m]["comm"] = comm_timing[m]
            out[m]["comm_exposed_ms_per_step"] = comm_timing[m]["comm_exposed_ms_per_step"]
    if world > 1 and "replicas" in results:
        # BASELINE configs[3] as the reference runs it (src/train_online.py:178-189: one sequence per process, no collective)
        out["replicas"]["parity_preserving"] = True
        out["cfg3_headline"] = {"replicas": out["replicas"]["value"] if head != "replicas" else out["value"],
                                "dp": out["value"] if head == "dp" else out.get("dp", {}).get("value"),
                                "unit": "frames/s",
                                "note": "replicas = the reference's own multi-device mapping (independent sequences, results "
                                        "identical to one GPU); dp = weak scaling with 5 micro-batches per rank and step"}
    if cold is not None:
        out["cold"] = {"value": args.steps / cold, "unit": "frames/s", "ms_per_step": cold / args.steps * 1000.0,
                       "note": f"the same {args.steps} steps behind {args.warmup} warm-up steps only, timed first (no preconditioning)"}
    if world == 1 and not args.no_variants and want("steps100") and args.steps != 100:
        # this file's default window (100 steps, 20 warm-up) beside the driver's --steps 20 --warmup 5: the same loop, the
        # per-call costs of a short window (0.25 ms of host set-up and tail) amortised
        saved_steps, saved_warm = args.steps, args.warmup
        args.steps, args.warmup = 100, 20
        try:
            e = timed("steps100")[0]
        finally:
            args.steps, args.warmup = saved_steps, saved_warm
        out["steps100"] = {"value": 100 / e, "unit": "frames/s", "ms_per_step": e / 100 * 1000.0, "steps": 100, "warmup": 20}
    if world == 1 and not args.no_variants and want("power"):
        # What the card reports while the step runs for a few seconds (amd-smi, read-only, four times a second from a child
        # process; none of it inside the headline's timed region): socket power against its limit and the GFX clocks.
        # Context (profiles/r04_lab_power_clocks.txt): the forward / data-gradient kernels ALONE sit at the 1400 W limit
        # with the clocks at 1.9-2.1 GHz, while the bare LDS-fed MFMA loop sustains 0.78 of peak at 1255 W.
        if power_sampler is not None:
            n_power = 4000
            open(power_sampler[1] + ".go", "w").close()  # wakes the sampler
            try:
                e = _with_steps(args, n_power, 10, lambda: timed("power")[0])
                t_end = time.time()
            finally:
                os.unlink(power_sampler[1] + ".go")
            power = _power_summary(power_sampler, t_end - e, t_end, n_power, e)
            if power is not None:
                out["power"] = power
    if world == 1 and not args.no_variants and want("group1"):
        # one frame per pass: the reference's own order, and the worst case of the shape bucketing
        e = timed("group1", env={"FOSVOS_MICROBATCH_GROUP": "1"})[0]
        probe = {role: _engine.STREAM_PROBE.get((dev_index, role)) for role in ("aux", "pass")}
        out["group1"] = {"value": args.steps / e, "unit": "frames/s", "ms_per_step": e / args.steps * 1000.0,
                         "note": "FOSVOS_MICROBATCH_GROUP=1: every micro-batch its own forward / backward pass",
                         # measured when the streams were created (engine.shared_stream): do the pass stream and the
                         # weight-gradient stream run beside the caller's stream and beside each other?
                         "pass_streams_overlap": (probe["pass"] or {}).get("overlaps_with"),
                         "aux_stream_overlap": (probe["aux"] or {}).get("overlaps_with"),
                         "stream_probe": probe}
    if world == 1 and not args.no_variants and want("mixed_scales"):
        # the reference's augmentation: a random scale per iteration, fixed seed; 20 frames per epoch
        import random
        rng = random.Random(1234)
        scales = [rng.choice((1.0, 0.8, 0.5)) for _ in range(20)]
        mixed = []
        for i, sc in enumerate(scales):
            hh, ww = int(H * sc), int(W * sc)  # cv2.resize(fx=fy=sc) sizes: 480x854, 384x683, 240x427
            im, g = make_frame(hh, ww, seed=1234, index=i)
            mixed.append({"image": im.unsqueeze(0).to(dev), "gt": g.unsqueeze(0).to(dev)})
        saved_steps, saved_warm = args.steps, args.warmup
        args.steps, args.warmup = 100, 20
        try:
            if os.environ.get("FOSVOS_BENCH_DEBUG"):
                st = torch.cuda.memory_stats()
                print("before mixed: reserved %.1f GB allocated %.1f GB device_allocs %d retries %d" % (
                    st["reserved_bytes.all.current"] / 2**30, st["allocated_bytes.all.current"] / 2**30,
                    st["num_device_alloc"], st["num_alloc_retries"]), file=sys.stderr)
            e = timed("mixed", precondition=40, loader=mixed)[0]
            if os.environ.get("FOSVOS_BENCH_DEBUG"):
                st = torch.cuda.memory_stats()
                print("after mixed: reserved %.1f GB allocated %.1f GB device_allocs %d device_frees %d retries %d host_lead %s" % (
                    st["reserved_bytes.all.current"] / 2**30, st["allocated_bytes.all.current"] / 2**30,
                    st["num_device_alloc"], st["num_device_free"], st["num_alloc_retries"], host_lead.get("mixed")), file=sys.stderr)
        finally:
            args.steps, args.warmup = saved_steps, saved_warm
        px = sum(b["image"].shape[2] * b["image"].shape[3] for b in mixed) / len(mixed)
        out["mixed_scales"] = {"value": 100 / e, "unit": "frames/s", "ms_per_step": e / 100 * 1000.0, "steps": 100,
                               "scales": {str(k): scales.count(k) for k in (1.0, 0.8, 0.5)},
                               "mean_pixels_per_frame": px, "full_frame_equivalents_per_s": 100 / e * px / (H * W),
                               "note": "frames drawn from {1.0, 0.8, 0.5} x 480x854 (seed 1234, 20 per epoch), bucketed by "
                                       "shape inside each accumulation cycle of 5 (src/dataloaders/custom_transforms.py:63-76)"}
        del mixed
    if world == 1 and not args.no_variants and want("davis_tree"):
        # The training half of train_and_test on a DAVIS-shaped tree ON DISK (one sequence, 480x854 JPEG + PNG mask written
        # here): the loader the script itself builds - io_helper.get_data_loader_train(root, 1, seq) - feeds _train, so the
        # rate INCLUDES the loader: the one-shot sample's six flip / scale variants resident on the device, drawn per epoch
        # with the reference pipeline's random numbers (dataloaders/resident.py).  `per_iteration_pipeline` = the same
        # loop behind the reference's own arrangement (DataLoader, one worker start + decode + resample per iteration).
        import tempfile
        import numpy as np
        from PIL import Image
        from util import io_helper
        root = tempfile.mkdtemp(prefix="fosvos_davis_")
        for sub in ("JPEGImages/480p/bench", "Annotations/480p/bench", "ImageSets/480p"):
            os.makedirs(os.path.join(root, sub))
        rs = np.random.RandomState(1234)
        yy, xx = np.mgrid[0:H, 0:W]
        lines = []
        for kf in range(2):
            m = (((yy - 0.45 * H) / (0.2 * H)) ** 2 + ((xx - (0.5 + 0.02 * kf) * W) / (0.16 * W)) ** 2 <= 1.0)
            fr = (rs.randint(0, 150, size=(H, W, 3)) + 100 * m[:, :, None]).astype(np.uint8)
            Image.fromarray(fr).save(os.path.join(root, "JPEGImages/480p/bench/%05d.jpg" % kf), quality=95)
            Image.fromarray((m * 255).astype(np.uint8)).save(os.path.join(root, "Annotations/480p/bench/%05d.png" % kf))
            lines.append("/JPEGImages/480p/bench/%05d.jpg /Annotations/480p/bench/%05d.png\n" % (kf, kf))
        for split in ("train", "val", "trainval"):
            with open(os.path.join(root, "ImageSets/480p", split + ".txt"), "w") as f:
                f.write("".join(lines))
        torch.manual_seed(1234)
        t0 = time.perf_counter()
        res_loader = io_helper.get_data_loader_train(root, 1, "bench")
        build_s = time.perf_counter() - t0
        saved_steps, saved_warm = args.steps, args.warmup
        args.steps, args.warmup = 100, 20
        try:
            e = timed("davis_tree", precondition=40, loader=res_loader)[0]
            draws = list(res_loader.draws[-100:])
            args.steps, args.warmup = 10, 2
            e_pipe = timed("davis_pipeline", precondition=0, loader=io_helper.get_data_loader_train(root, 1, "bench", resident=False))[0]
        finally:
            args.steps, args.warmup = saved_steps, saved_warm
        px = sum(res_loader.variants[d]["image"].shape[2] * res_loader.variants[d]["image"].shape[3] for d in draws) / len(draws)
        out["davis_tree"] = {"value": 100 / e, "unit": "frames/s", "ms_per_step": e / 100 * 1000.0, "steps": 100,
                             "loader": type(res_loader).__name__, "loader_build_s": build_s,
                             "mean_pixels_per_frame": px, "full_frame_equivalents_per_s": 100 / e * px / (H * W),
                             "per_iteration_pipeline": {"value": 10 / e_pipe, "unit": "frames/s", "steps": 10,
                                                        "ms_per_step": e_pipe / 10 * 1000.0},
                             "note": "train_online._train fed by the loader train_and_test builds for a sequence run on a "
                                     "DAVIS tree on disk (src/util/io_helper.py:62-70, src/dataloaders/davis_2016.py:72-83)"}
        import shutil
        shutil.rmtree(root, ignore_errors=True)
    if world == 1 and not args.no_variants and want("offline"):
        out["offline"] = offline_config(args, dev, make_frame, barrier)

    if rank == 0 and not args.no_roofline:
        # rank-0-only pass, so no collective: the single-process loop (what every replica runs; the dp loop differs only by
        # the all-reduce).  Same shipped execution as the timed region - native layer loop, two streams, deferred join -
        # with the library's event pair around every launch on its own stream.
        train_online.data_parallel = False
        n_prof = max(AVG_GRAD_EVERY_N, args.prof_steps // AVG_GRAD_EVERY_N * AVG_GRAD_EVERY_N)
        with LaunchProfile(dev_index) as prof:
            train_online._train(prov, batch, opt, _NullWriter(), "bench", 0, n_prof, AVG_GRAD_EVERY_N, 10 ** 9)
        by_kernel = {}
        for name, a in sorted(prof.records.items(), key=lambda kv: -kv[1]["ms"]):
            by_kernel[name] = {
                "launches_per_step": a["launches"] / n_prof,
                "ms_per_step": round(a["ms"] / n_prof, 4),
                "avg_us_per_launch": round(a["ms"] / a["launches"] * 1000.0, 2),
                "tflops": round(a["flops"] / a["ms"] / 1e9, 2) if a["flops"] else None,
            }
        conv = [a for k, a in prof.records.items() if k.startswith(CONV_KERNEL_PREFIXES)]
        conv_ms = sum(a["ms"] for a in conv) / n_prof
        conv_calls = sum(a["launches"] for a in conv) / n_prof
        conv_flop = sum(a["flops"] for a in conv) / n_prof
        fam_achieved = conv_flop / (conv_ms * 1e-3) / 1e12
        # The dominant kernel = the single kernel NAME (one instantiation, as rocprofv3 prints it) with the most time per step:
        # its time per step cannot exceed the step, whatever overlaps with it.  Beside it: `template` (all instantiations of
        # the kernel template with the most time per step - what this field carried in round 3), `family` (all conv MFMA
        # kernels), `alone`, `step_wall_frac`.
        dom_name, dom = max(((k, a) for k, a in prof.records.items() if k.startswith(CONV_KERNEL_PREFIXES)),
                            key=lambda kv: kv[1]["ms"])
        achieved = dom["flops"] / dom["ms"] / 1e9
        groups = {}
        for k, a in prof.records.items():
            if k.startswith(CONV_KERNEL_PREFIXES):
                g = groups.setdefault(k.split("<")[0], {"ms": 0.0, "flops": 0.0, "launches": 0, "names": []})
                g["ms"] += a["ms"]; g["flops"] += a["flops"]; g["launches"] += a["launches"]; g["names"].append(k)
        tpl_name, tpl = max(groups.items(), key=lambda kv: kv[1]["ms"])
        # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this process); null when no
        # summary is present or the kernel is not in it
        traffic, tpl_traffic, fam_traffic, traffic_note = None, None, None, "no profiles/*_pmc_traffic.json present"
        try:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
            if cands:
                pm = json.load(open(cands[-1]))
                fam_traffic = pm["summary"]["conv_mfma_family"]["hbm_bytes_per_launch"]

                def per_launch(names):
                    tb, tl = 0.0, 0
                    for k, v in pm.get("per_kernel", {}).items():
                        if any(k.endswith(n) for n in names):
                            tb += (v["hbm_read_bytes_per_launch"] + v["hbm_write_bytes_per_launch"]) * v["launches"]
                            tl += v["launches"]
                    return tb / tl if tl else None
                traffic, tpl_traffic = per_launch([dom_name]), per_launch(tpl["names"])
                traffic_note = (f"HBM bytes/launch of this kernel from {os.path.relpath(cands[-1], ROOT)}: {pm['source']}; "
                                f"{pm['corrections']}")
        except Exception as e:  # a malformed summary must not break the bench line
            traffic_note = f"could not read PMC summary: {e}"
        out["roofline"] = {
            "bound": "mfma",
            "kernel": dom_name + " (the kernel with the most time per step; bf16 operands, fp32 accumulate)",
            "achieved": achieved,
            "peak": MFMA_BF16_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / MFMA_BF16_PEAK_TFLOPS,
            "traffic": traffic,
            "traffic_note": traffic_note,
            "measured": f"HIP event pairs on each kernel's launch stream, {n_prof} steps of the shipped two-stream execution "
                        f"right behind the timed region.  Kernels overlap, as under rocprofv3: the forward pass runs as two "
                        f"chains of frames on the two streams, the backward pass as a data-gradient and a weight-gradient "
                        f"stream, so a kernel's duration is that of a kernel sharing the chip; `alone` = the same kernels with "
                        f"the chip to themselves, `step_wall_frac` = the step's FLOPs over its wall time",
            "launches_per_step": dom["launches"] / n_prof,
            "ms_per_step": dom["ms"] / n_prof,
            "algorithmic_gflop_per_launch": dom["flops"] / dom["launches"] / 1e9,
            "avg_launch_us": dom["ms"] / dom["launches"] * 1000.0,
            # the kernel TEMPLATE with the most time per step, all its instantiations (their launches overlap each other, so
            # this sum may exceed the step)
            "template": {"kernel": tpl_name, "instantiations": sorted(n[len(tpl_name):] for n in tpl["names"]),
                         "achieved": tpl["flops"] / tpl["ms"] / 1e9, "frac": tpl["flops"] / tpl["ms"] / 1e9 / MFMA_BF16_PEAK_TFLOPS,
                         "launches_per_step": tpl["launches"] / n_prof, "ms_per_step": tpl["ms"] / n_prof,
                         "avg_launch_us": tpl["ms"] / tpl["launches"] * 1000.0, "traffic": tpl_traffic},
            # all conv MFMA kernels together
            "family": {"achieved": fam_achieved, "frac": fam_achieved / MFMA_BF16_PEAK_TFLOPS,
                       "launches_per_step": conv_calls, "algorithmic_gflop_per_step": conv_flop / 1e9,
                       "avg_launch_us": conv_ms / conv_calls * 1000.0, "traffic": fam_traffic},
            "device_ms_per_step_all_kernels": sum(a["ms"] for a in prof.records.values()) / n_prof,
            "non_mfma_ms_per_step": sum(a["ms"] for k, a in prof.records.items() if not k.startswith(CONV_KERNEL_PREFIXES)) / n_prof,
            "launches_per_step_all_kernels": sum(a["launches"] for a in prof.records.values()) / n_prof,
            # the same FLOPs over the WALL time of a step of the timed region (everything included: the streams overlap,
            # so the kernel durations above add up to more than the step; the non-MFMA kernels, launch gaps and the optimizer
            # step are in it too): what the whole step sustains, next to what its kernels sustain while they run
            "step_wall_tflops": conv_flop / 1e12 / (elapsed / args.steps),
            "step_wall_frac": conv_flop / 1e12 / (elapsed / args.steps) / MFMA_BF16_PEAK_TFLOPS,
            "by_kernel": by_kernel,
        }
        if not args.no_alone:
            out["roofline"]["alone"] = alone_rates(dev, AVG_GRAD_EVERY_N, H, W)
    if rank == 0 and not args.no_infer:
        # f1: the reference's eval_speeds protocol on 480x854 frames, all five logit maps computed
        from util import experiment_helper, io_helper
        # frames materialised up front: generating a synthetic frame on the host takes ~10 ms, long enough for the idle GPU
        # to drop its clocks between two timed forwards (the bracket then measures the clock ramp: 0.8-1.8 ms per frame)
        loader = list(io_helper.get_data_loader_test(None, 1, "bench", synthetic=(H, W), n_frames=6))
        import gc
        gc.collect()  # tensors of the training passes die here, not inside a timed bracket (seen: one 79 ms sample in 50)
        torch.cuda.synchronize()
        sec = experiment_helper.test(prov, loader, os.path.join(ROOT, "gpurun_out", "bench_infer"), False, True, seq_name="bench")
        ev = experiment_helper.last_eval
        if os.environ.get("FOSVOS_BENCH_DEBUG"):
            print("infer times ms:", ["%.2f" % (t * 1e3) for t in ev["times"]], file=sys.stderr)
        med = sorted(ev["times"])[len(ev["times"]) // 2]
        out["infer"] = {"infer_ms_per_frame": sec * 1000.0, "median_ms_per_frame": med * 1000.0, "frames_per_s": 1.0 / sec,
                        "frame": [H, W], "outputs": 5,
                        "protocol": f"net.forward between device syncs, {ev['n_runs']} passes x {len(loader)} frames, first "
                                    f"frame of each pass dropped ({len(ev['times'])} samples; host-to-device copy of the "
                                    f"frame outside the bracket), src/util/experiment_helper.py:29-53",
                        "reference_published_s_per_frame": 0.08083}
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
