#!/usr/bin/env python3
"""Single-frame forward at 480x854 (GPU box): wall time per frame under the eval_speeds protocol next to the per-kernel event
durations of the same launches (fosvos_hip.LaunchProfile) - how much of a frame is kernels and how much the gaps between
dependent launches."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))
import fosvos_amd  # noqa
from dataloaders.synthetic import make_frame
from fosvos_hip import LaunchProfile
from networks.osvos_vgg import OSVOS_VGG

torch.manual_seed(0)
net = OSVOS_VGG(pretrained=0).cuda()
for n_, p in net.named_parameters():
    if 'stages' in n_ and 'weight' in n_: torch.nn.init.kaiming_normal_(p)
img, gt = make_frame(480, 854)
x = img.unsqueeze(0).cuda()
with torch.no_grad():
    for _ in range(20): net.forward(x)
    ts = []
    for _ in range(40):
        torch.cuda.synchronize(); t0 = time.perf_counter(); net.forward(x); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print("protocol: median %.3f ms, min %.3f ms" % (ts[20] * 1e3, ts[0] * 1e3))
    n = 20
    with LaunchProfile(0) as prof:
        for _ in range(n): net.forward(x)
    tot = 0.0
    for k, a in sorted(prof.records.items(), key=lambda kv: -kv[1]["ms"]):
        print("  %-62s n=%4.1f  us/frame=%7.1f  avg_us=%7.1f" % (k[:62], a["launches"] / n, a["ms"] / n * 1e3, a["ms"] / a["launches"] * 1e3))
        tot += a["ms"] / n
    print("kernel time per frame: %.3f ms, launches per frame: %.1f" % (tot, sum(a["launches"] for a in prof.records.values()) / n))
