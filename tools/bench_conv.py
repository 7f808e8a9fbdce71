#!/usr/bin/env python3
"""Per-layer microbenchmark of the conv kernels (GPU box): times fwd / dgrad / wgrad of the 13 backbone
convs + 4 side_prep convs at 1x480x854 with HIP events, prints TFLOP/s per call.  Used for A/B kernel work and
under rocprofv3 --pmc.   usage: bench_conv.py [reps] [filter-substring] [ops=fwd,dgrad,wgrad]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))
from fosvos_hip import ops  # noqa: E402

LAYERS = [  # name, H, W, Ci, Co
    ("conv1_2", 480, 854, 64, 64), ("conv2_1", 240, 427, 64, 128), ("conv2_2", 240, 427, 128, 128),
    ("conv3_1", 120, 214, 128, 256), ("conv3_2", 120, 214, 256, 256), ("conv4_1", 60, 107, 256, 512),
    ("conv4_2", 60, 107, 512, 512), ("conv5_1", 30, 54, 512, 512),
    ("side1", 240, 427, 128, 16), ("side2", 120, 214, 256, 16), ("side3", 60, 107, 512, 16), ("side4", 30, 54, 512, 16),
]


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    which = (sys.argv[3] if len(sys.argv) > 3 else "fwd,dgrad,wgrad").split(",")
    dev = "cuda:0"
    nb = int(os.environ.get("BENCH_CONV_N", "1"))  # frames per launch
    g = torch.Generator(device=dev).manual_seed(0)
    print(f"{'layer':9s} {'op':6s} {'us':>9s} {'TFLOP/s':>9s}")
    tot = {}
    for name, H, W, ci, co in LAYERS:
        if filt and filt not in name:
            continue
        x = torch.randn(nb, H, W, ci, device=dev, generator=g).to(torch.bfloat16)
        w = torch.randn(co, ci, 3, 3, device=dev, generator=g) * (2.0 / (9 * ci)) ** 0.5
        b = torch.randn(co, device=dev, generator=g) * 0.1
        wf, wd = ops.pack_conv3x3_weights(w)
        cy = (co + 31) // 32 * 32
        dy = torch.randn(nb, H, W, cy, device=dev, generator=g).to(torch.bfloat16)
        if cy != co:
            dy[..., co:] = 0
        if os.environ.get("BENCH_CONV_ZERO") == "1":  # DVFS check: the same instruction stream on all-zero operands
            x.zero_()
            dy.zero_()
        flop = 2.0 * nb * H * W * 9 * ci * co
        side = co == 16
        res = {}
        if "fwd" in which:
            res["fwd"] = timeit(lambda: ops.conv3x3_fwd(x, wf, b, ci, co, relu=not side, out_f32=side), reps)
        if "dgrad" in which:
            res["dgrad"] = timeit(lambda: ops.conv3x3_dgrad(dy, wd, ci, co, relu_src=x), reps)
        if "wgrad" in which:
            res["wgrad"] = timeit(lambda: ops.conv3x3_wgrad(x, dy, ci, co), reps)
        for op, us in res.items():
            print(f"{name:9s} {op:6s} {us:9.1f} {flop / us / 1e6:9.1f}")
            tot[op] = tot.get(op, 0.0) + us
    print("sum us:", {k: round(v, 1) for k, v in tot.items()})


if __name__ == "__main__":
    main()
