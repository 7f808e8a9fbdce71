#!/usr/bin/env python3
"""Stand-alone per-kernel timing of the weight-gradient path (GPU box): every layer shape of the 480x854 step, kernels
alone on the chip, durations from the library's own launch profiler (fosvos_profile_start/stop).
usage: wgrad_lab.py [reps] [filter]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))
from fosvos_hip import LaunchProfile, ops  # noqa: E402

LAYERS = [  # name, H, W, Ci, Co
    ("conv1_2", 480, 854, 64, 64), ("conv2_1", 240, 427, 64, 128), ("conv2_2", 240, 427, 128, 128),
    ("conv3_1", 120, 214, 128, 256), ("conv3_2", 120, 214, 256, 256), ("conv4_1", 60, 107, 256, 512),
    ("conv4_2", 60, 107, 512, 512), ("conv5_1", 30, 54, 512, 512),
    ("side1", 240, 427, 128, 16), ("side2", 120, 214, 256, 16), ("side3", 60, 107, 512, 16), ("side4", 30, 54, 512, 16),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    print(f"{'layer':9s} {'kernel':28s} {'us':>9s} {'TFLOP/s':>9s}")
    tot = 0.0
    for name, H, W, ci, co in LAYERS:
        if filt and filt not in name:
            continue
        x = torch.randn(1, H, W, ci, device=dev, generator=g).to(torch.bfloat16)
        cy = (co + 31) // 32 * 32
        dy = torch.randn(1, H, W, cy, device=dev, generator=g).to(torch.bfloat16)
        if cy != co:
            dy[..., co:] = 0
        for _ in range(3):
            ops.conv3x3_wgrad(x, dy, ci, co)
        torch.cuda.synchronize()
        with LaunchProfile(0) as prof:
            for _ in range(reps):
                ops.conv3x3_wgrad(x, dy, ci, co)
        for k, a in prof.records.items():
            us = a["ms"] / a["launches"] * 1e3
            tf = a["flops"] / a["ms"] / 1e9 if a["flops"] else 0.0
            print(f"{name:9s} {k:28s} {us:9.1f} {tf:9.1f}")
            tot += us * (3 if name == "conv5_1" else 2 if name in ("conv3_2", "conv4_2") else 1)
    print(f"sum over the step's 16 layers: {tot:.1f} us")


if __name__ == "__main__":
    main()
