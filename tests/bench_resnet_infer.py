"""OSVOS_RESNET inference at 1920x1080 on one MI355X (BASELINE.json configs[4]): ms/frame for a few
(version, scale_down_exponent) pairs, a per-kernel breakdown from HIP events, and a parity spot check against the CPU
oracle at a small size.  Not the headline metric - bench.py stays on the OSVOS_VGG fine-tune.
usage: python tests/bench_resnet_infer.py [--json out.json] [--cpu]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))

from fosvos_hip import ops, resnet_engine  # noqa: E402
from networks.osvos_resnet import OSVOS_RESNET  # noqa: E402
from oracle import osvos_resnet_ref as R  # noqa: E402  (checker + CPU baseline leg only)

ap = argparse.ArgumentParser()
ap.add_argument("--json", default=None)
ap.add_argument("--cpu", action="store_true", help="also time the fp32 oracle on the host cores (one frame)")
ap.add_argument("--hw", default="1080x1920")
ap.add_argument("--detail", action="store_true", help="one line per conv shape")
ap.add_argument("--native-only", action="store_true", help="skip the parity check and the per-kernel pass (clean rocprofv3 traces)")
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--nets", default="18:0,18:1,18:2,18:3,34:2")
args = ap.parse_args()
H, W = (int(v) for v in args.hw.split("x"))
dev = "cuda:0"
results = []
for version, e in [tuple(int(v) for v in t.split(":")) for t in args.nets.split(",")]:
    sd = R.make_state_dict(version, e, seed=1)
    net = OSVOS_RESNET(pretrained=False, version=version, scale_down_exponent=e)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    # parity spot check at a size the oracle does in a second
    rel = float("nan")
    if not args.native_only:
        xs = 50.0 * torch.randn(1, 3, 96, 160, generator=torch.Generator().manual_seed(3))
        got = net(xs.to(dev))[-1].cpu()
        ref = R.forward(sd, xs)[-1]
        rel = ((got - ref).abs().max() / ref.abs().max()).item()
    x = (50.0 * torch.randn(1, 3, H, W, generator=torch.Generator().manual_seed(4))).to(dev)
    # the card idles (and clocks down) while the host builds the net and runs the oracle: warm up for 0.3 s, then time
    # enough frames that the region is ~0.2 s or more
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(10):
            net(x)
        torch.cuda.synchronize()
    reps = args.reps
    t0 = time.perf_counter()
    for _ in range(reps):
        out = net(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    agg = {}
    if not args.native_only:
        for _ in range(10):
            resnet_engine.forward_ops(net, net._plan, x)
        prof = ops.OpProfiler(detail=args.detail)
        ops.set_profiler(prof)
        for _ in range(3):
            resnet_engine.forward_ops(net, net._plan, x)  # same kernels, one C-ABI call each, so that they can be timed apart
        agg = prof.summary()
        ops.set_profiler(None)
    by = {k: {"calls_per_frame": v["calls"] / 3, "ms_per_frame": v["ms"] / 3,
              "tflops": (v["flops"] / v["ms"] * 1e-9) if v["flops"] else None, "gbs": v["bytes"] / v["ms"] * 1e-6}
          for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
    gflop = sum(v["flops"] for v in agg.values()) / 3 * 1e-9
    act_bytes = sum(v["bytes"] for v in agg.values()) / 3
    row = {"version": version, "scale_down_exponent": e, "frame": [H, W], "ms_per_frame": ms, "frames_per_s": 1e3 / ms,
           "gflop_per_frame_issued": gflop, "algorithmic_mb_per_frame": act_bytes * 1e-6,
           "algorithmic_gbs": act_bytes / ms * 1e-6, "hbm_frac_of_8tbs": act_bytes / ms * 1e-6 / 8000.0, "device_ms_per_frame": sum(v["ms"] for v in agg.values()) / 3,
           "logit_err_vs_oracle_96x160": rel, "by_kernel": by}
    if args.cpu and e >= 2:
        xc = x.cpu()
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        t0 = time.perf_counter()
        with torch.no_grad():
            R.forward(sd, xc)
        row["cpu_oracle_ms_per_frame"] = (time.perf_counter() - t0) * 1e3
        row["cpu_threads"] = torch.get_num_threads()
    results.append(row)
    print("resnet%d e=%d %dx%d: %.3f ms/frame (%.0f fps), device %.3f ms, %.1f GFLOP issued/frame (zero-padded channels included), err %.2e%s" % (
        version, e, H, W, ms, 1e3 / ms, row["device_ms_per_frame"], gflop, rel,
        (", cpu oracle %.0f ms" % row["cpu_oracle_ms_per_frame"]) if "cpu_oracle_ms_per_frame" in row else ""))
    for k, v in by.items():
        print("    %-34s %5.1f calls %8.3f ms  %s  %7.0f GB/s" % (
            k, v["calls_per_frame"], v["ms_per_frame"], ("%6.1f TFLOP/s" % v["tflops"]) if v["tflops"] else " " * 14, v["gbs"]))
if args.json:
    with open(args.json, "w") as f:
        json.dump(results, f, indent=1)
