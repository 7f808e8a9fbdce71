#!/usr/bin/env python3
"""Phase stamps of the persistent forward conv kernel (lab build only): per super-step, when it began, when the work
was done (before the barrier), and when the barrier let it go.  usage: FOSVOS_HIP_LIB=.../libfosvos_hip_lab.so pp_stamp_lab.py [N H W Ci Co]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))
import fosvos_hip  # noqa: E402
from fosvos_hip import ops  # noqa: E402


def main():
    n, h, w, ci, co = (int(v) for v in sys.argv[1:6]) if len(sys.argv) >= 6 else (5, 480, 854, 64, 64)
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(n, h, w, ci, device=dev, generator=g).to(torch.bfloat16)
    wt = torch.randn(co, ci, 3, 3, device=dev, generator=g) * (2.0 / (9 * ci)) ** 0.5
    b = torch.randn(co, device=dev, generator=g) * 0.1
    wf, _ = ops.pack_conv3x3_weights(wt)
    stamps = torch.zeros(8 * 2 * 512, dtype=torch.int64, device=dev)
    L = fosvos_hip.lib()
    L.fosvos_lab_set_pp_stamps.argtypes = [ctypes.c_void_p]
    for _ in range(30):  # warm: clocks up
        ops.conv3x3_fwd_pool(x, wf, b, ci, co, relu=True)
    torch.cuda.synchronize()
    L.fosvos_lab_set_pp_stamps(stamps.data_ptr())
    ops.conv3x3_fwd_pool(x, wf, b, ci, co, relu=True)
    torch.cuda.synchronize()
    L.fosvos_lab_set_pp_stamps(None)
    s = stamps.cpu().numpy().reshape(8, 2, 512)
    nc = ci // 32
    for wg in (0, 3):
        t0 = s[wg, 0, 0]
        nst = int((s[wg, 0, :256] > 0).sum())
        K = (nst - 1) // 4
        print(f"workgroup {wg}: {K} super-steps, {nc} chunks per tile (lockstep schedule); clocks")
        print("   k | step length | g0: requests, taps+stores, dma wait | g1: taps+stores+requests, dma wait")
        rows = []
        for k in range(K):
            a = [int(v - t0) for v in s[wg, 0, 4 * k:4 * k + 5]]
            b = [int(v - t0) for v in s[wg, 1, 4 * k:4 * k + 4]]
            rows.append((a[4] - a[0], a[1] - a[0], a[2] - a[1], a[3] - a[2], b[2] - b[0], b[3] - b[2]))
            if k < 12 or k >= K - 3:
                print(f"{k:4d} | {a[4] - a[0]:8d} | {a[1] - a[0]:6d} {a[2] - a[1]:6d} {a[3] - a[2]:6d} | {b[2] - b[0]:6d} {b[3] - b[2]:6d}")
        r = np.array(rows[2:-2])
        print("   mean (steady state):", " ".join(f"{v:8.0f}" for v in r.mean(axis=0)))
        print(f"   total {int(s[wg, 0, 4 * K] - t0)} clocks = {K} x {int(s[wg, 0, 4 * K] - t0) / max(K, 1):.0f}")


if __name__ == "__main__":
    main()
