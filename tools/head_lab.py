"""Time fosvos_deconv_head_fwd alone at 1080p, with and without the four side outputs.  usage: python tools/head_lab.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))
from fosvos_hip import ops
H, W = 1080, 1920
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
sizes = [(270, 480), (135, 240), (68, 120), (34, 60)]
side = [torch.randn(1, a, b, 16, generator=g).to(dev) for a, b in sizes]
filt = [torch.randn(8 << s, 8 << s, 16, generator=g).to(dev) for s in range(4)]
filt1 = [torch.randn(8 << s, 8 << s, generator=g).to(dev) for s in range(4)]
dsn_w = torch.randn(4, 16, generator=g).to(dev); dsn_b = torch.randn(4, generator=g).to(dev); fb = torch.randn(1, generator=g).to(dev)
for with_so in (True, False):
    for _ in range(50):
        ops.deconv_head_fwd(side, [4, 8, 16, 32], filt, filt1, dsn_w, dsn_b, fb, H, W, with_so)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        ops.deconv_head_fwd(side, [4, 8, 16, 32], filt, filt1, dsn_w, dsn_b, fb, H, W, with_so)
    e1.record(); torch.cuda.synchronize()
    print("side outputs %s: %.1f us per call" % (with_so, e0.elapsed_time(e1) / 200 * 1e3))
