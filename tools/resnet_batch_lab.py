"""Frames/s of the native OSVOS_RESNET forward against the batch size (the thin nets are bound by per-launch cost, which a
batch amortises).  usage: python tools/resnet_batch_lab.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))
from networks.osvos_resnet import OSVOS_RESNET
dev = "cuda:0"
for version, e in ((18, 2), (18, 0)):
    torch.manual_seed(1)
    net = OSVOS_RESNET(pretrained=False, version=version, scale_down_exponent=e)
    for m in net.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    net = net.to(dev).eval()
    for n in (1, 2, 4, 8):
        x = (50.0 * torch.randn(n, 3, 1080, 1920, generator=torch.Generator().manual_seed(4))).to(dev)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            for _ in range(5): net(x)
            torch.cuda.synchronize()
        reps = max(20, 200 // n)
        t0 = time.perf_counter()
        for _ in range(reps): net(x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        print("resnet%d e=%d batch %d: %.3f ms per forward, %.3f ms per frame, %.0f frames/s" % (version, e, n, ms, ms / n, n * 1e3 / ms))
