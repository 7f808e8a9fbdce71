"""Time fosvos_conv2d_fwd (the vector-ALU direct conv) alone on a few thin-layer shapes; run under rocprofv3 --pmc for
counters.  usage: python tools/conv2d_lab.py [ci:co:h:w:k:stride ...]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))
from fosvos_hip import ops
dev = "cuda:0"
shapes = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(16, 16, 270, 480, 3, 1), (32, 32, 135, 240, 3, 1), (8, 8, 270, 480, 3, 1), (16, 32, 270, 480, 3, 2)]
g = torch.Generator().manual_seed(0)
for ci, co, h, w, k, st in shapes:
    x = torch.randn(1, h, w, (ci + 7) // 8 * 8, generator=g).to(torch.bfloat16).to(dev)
    wt = (torch.randn(co, ci, k, k, generator=g) * 0.1).to(dev)
    packed, bias = ops.pack_conv2d_bn(wt, None, None)
    for _ in range(30):
        ops.conv2d_fwd(x, packed, bias, ci, co, k, st, True)
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        y = ops.conv2d_fwd(x, packed, bias, ci, co, k, st, True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    fl = 2.0 * y.shape[1] * y.shape[2] * k * k * ci * co
    print("conv%dx%d s%d %d->%d @%dx%d: %.1f us  %.1f TFLOP/s" % (k, k, st, ci, co, h, w, us, fl / us * 1e-6))
