"""Host-side cost of one OSVOS_RESNET.forward call against the device time per frame: enqueue-only time (no sync inside the
timed region, a long queue already pending) vs synchronous throughput.  usage: python tools/resnet_host_lab.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))
from networks.osvos_resnet import OSVOS_RESNET
from fosvos_hip.resnet_engine import ResnetPlan
dev = "cuda:0"
for version, e in ((18, 2), (18, 3), (34, 2)):
    torch.manual_seed(1)
    net = OSVOS_RESNET(pretrained=False, version=version, scale_down_exponent=e).to(dev).eval()
    x = torch.randn(1, 3, 1080, 1920, device=dev)
    for _ in range(50): net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): net(x)
    t_enq = (time.perf_counter() - t0) / 300 * 1e3
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 300 * 1e3
    t0 = time.perf_counter()
    for _ in range(300): ResnetPlan._signature(net)
    t_sig = (time.perf_counter() - t0) / 300 * 1e3
    print("resnet%d e=%d: enqueue %.3f ms per call, with final sync %.3f ms per call, signature alone %.3f ms" % (version, e, t_enq, t_all, t_sig))
