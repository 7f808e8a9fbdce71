#!/usr/bin/env python3
"""Lab tool (not product): which weight does the 3x3 igemm apply where?

One bright pixel in one input channel; the weights carry small integer codes that bf16 holds exactly.  The 3x3
neighbourhood of the bright pixel in output channel co then shows w[co, c0, tap] for the nine taps, so a staging bug in
the weight image (wrong tap, wrong k-group, wrong channel, stale slot) is readable from the decoded codes.

    python tools/ring_probe.py [Cin] [Cout]        # FOSVOS_IGEMM_RING=0/1 selects the form under test
"""
import sys

import torch

sys.path.insert(0, ".")
from fosvos_amd.fosvos_hip import ops  # noqa: E402

DEV = torch.device("cuda:0")


def main():
    ci = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    co = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    n, h, w = 1, 240, 427
    print("plan", ops.conv3x3_plan(n, h, w, ci, co))
    t = torch.arange(9).view(1, 1, 3, 3).float()
    cof = torch.arange(co).view(co, 1, 1, 1).float()
    cif = torch.arange(ci).view(1, ci, 1, 1).float()
    # code A: tap + 10 (co % 16) + 1 (<= 160), code B: ci % 32 + 32 (co // 16 % 4) + 1, code C: ci // 32 + 4 * (co // 64) + 1
    codes = {
        "tap/co16": (t + 10 * (cof % 16) + 1 + 0 * cif),
        "ci32/cogrp": (0 * t + (cif % 32) + 32 * ((cof // 16) % 4) + 1),
        "chunk/coblk": (0 * t + (cif // 32) + 4 * (cof // 64) + 1),
    }
    py, px = 100, 200
    for name, wt in codes.items():
        wt = wt.contiguous()
        wf, _ = ops.pack_conv3x3_weights(wt.to(DEV))
        bad = 0
        shown = 0
        for c0 in range(ci):
            x = torch.zeros(n, h, w, ci, dtype=torch.bfloat16, device=DEV)
            x[0, py, px, c0] = 1.0
            y = ops.conv3x3_fwd(x, wf, None, ci, co, relu=False)
            torch.cuda.synchronize()
            # out[co, py - (ky - 1), px - (kx - 1)] = w[co, c0, ky, kx]
            got = torch.empty(co, 3, 3)
            for ky in range(3):
                for kx in range(3):
                    got[:, ky, kx] = y[0, py - (ky - 1), px - (kx - 1), :].float().cpu()
            want = wt[:, c0]
            miss = (got != want)
            bad += int(miss.sum())
            if miss.any() and shown < 6:
                shown += 1
                idx = miss.nonzero()[:6]
                for (o, ky, kx) in idx.tolist():
                    print(f"  [{name}] c0={c0} co={o} tap=({ky},{kx}) want {want[o, ky, kx].item():.0f} got {got[o, ky, kx].item():.0f}")
            # anything outside the neighbourhood must be zero
            y[0, py - 1:py + 2, px - 1:px + 2, :] = 0
            stray = int((y != 0).sum())
            if stray:
                print(f"  [{name}] c0={c0}: {stray} stray non-zero outputs, e.g. {(y != 0).nonzero()[:4].tolist()}")
        print(f"{name}: {bad} wrong of {ci * co * 9}")


if __name__ == "__main__":
    main()
