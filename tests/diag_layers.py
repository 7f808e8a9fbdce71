#!/usr/bin/env python3
"""Diagnostic (GPU box): per-layer disagreement of the HIP forward/backward with the fp32 oracle and the
bf16-emulating oracle.  Not a test; prints a table."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))
from oracle import osvos_ref as O  # noqa: E402
from fosvos_hip import engine  # noqa: E402
from networks.osvos_vgg import OSVOS_VGG  # noqa: E402


def oracle_acts(sd, x, emu):
    rb = O._RoundBoth.apply if emu else (lambda t: t)
    rv = O._RoundValue.apply if emu else (lambda t: t)
    acts, sides = [], []
    h = x
    for s, chans in enumerate(O.STAGE_CHANNELS):
        if s > 0:
            h = F.max_pool2d(h, 2, 2, ceil_mode=True)
        for j in range(len(chans)):
            m = O.conv_module_index(s, j)
            w = sd[f"stages.{s}.{m}.weight"]
            if not (s == 0 and j == 0):
                w = rv(w)
            h = rb(F.relu(F.conv2d(h, w, sd[f"stages.{s}.{m}.bias"], padding=1)))
            acts.append(h)
        if s > 0:
            sides.append(F.conv2d(h, rv(sd[f"side_prep.{s-1}.weight"]), sd[f"side_prep.{s-1}.bias"], padding=1))
    return acts, sides


def main():
    n, h, w, seed = 2, 61, 107, 4
    if len(sys.argv) > 3:
        n, h, w = (int(v) for v in sys.argv[1:4])
    sd = O.make_state_dict(seed)
    x, gt = O.synthetic_frame(n, h, w, seed=100 + seed)
    net = OSVOS_VGG(pretrained=0)
    net.load_state_dict(sd)
    net = net.cuda()
    P = dict(net.named_parameters())
    with torch.no_grad():
        outs, sv = engine.forward(P, net._packs, x.cuda(), keep=True)
        a32, s32 = oracle_acts(sd, x, False)
        aem, sem = oracle_acts(sd, x, True)
    print(f"{'layer':14s} {'max|ref|':>10s} {'hip-emu/max':>12s} {'hip-fp32/max':>12s} {'emu-fp32/max':>12s} {'mism frac(hip,emu)':>18s}")
    for i, y in enumerate(sv.conv_out):
        hip = y.float().permute(0, 3, 1, 2).cpu()
        m = a32[i].abs().max().item()
        mism = (hip != aem[i]).float().mean().item()
        print(f"conv{i:<10d} {m:10.3f} {(hip-aem[i]).abs().max().item()/m:12.3e} {(hip-a32[i]).abs().max().item()/m:12.3e} "
              f"{(aem[i]-a32[i]).abs().max().item()/m:12.3e} {mism:18.5f}")
    for i, y in enumerate(sv.side):
        hip = y.permute(0, 3, 1, 2).cpu()
        m = s32[i].abs().max().item()
        print(f"side{i:<10d} {m:10.3f} {(hip-sem[i]).abs().max().item()/m:12.3e} {(hip-s32[i]).abs().max().item()/m:12.3e} "
              f"{(sem[i]-s32[i]).abs().max().item()/m:12.3e}")
    o32 = O.forward(sd, x)
    oem = O.forward(sd, x, emulate_bf16=True)
    for i in range(5):
        hip = outs[i].cpu()
        m = o32[i].abs().max().item()
        l_h = O.cbce_loss(hip, gt, False).item()
        l_e = O.cbce_loss(oem[i], gt, False).item()
        l_f = O.cbce_loss(o32[i], gt, False).item()
        print(f"out{i:<11d} {m:10.3f} {(hip-oem[i]).abs().max().item()/m:12.3e} {(hip-o32[i]).abs().max().item()/m:12.3e} "
              f"{(oem[i]-o32[i]).abs().max().item()/m:12.3e}   loss hip/emu/fp32 = {l_h:.2f} / {l_e:.2f} / {l_f:.2f}")


if __name__ == "__main__":
    main()
