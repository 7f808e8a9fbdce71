"""Does replaying the native ResNet forward as a HIP graph shorten the per-kernel gaps?  usage: python tools/resnet_graph_lab.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))
from networks.osvos_resnet import OSVOS_RESNET
dev = "cuda:0"
for version, e in ((18, 2), (18, 3), (18, 0)):
    torch.manual_seed(1)
    net = OSVOS_RESNET(pretrained=False, version=version, scale_down_exponent=e)
    for m in net.modules():  # alive activations: He-scaled convs, non-trivial BatchNorm statistics
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight)
        elif isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    net = net.to(dev).eval()
    x = (50.0 * torch.randn(1, 3, 1080, 1920, generator=torch.Generator().manual_seed(4))).to(dev)
    def timeit(fn, reps=300):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            for _ in range(10): fn()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    eager = timeit(lambda: net(x))
    ref = net(x)[-1].clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): net(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = net(x)
    g.replay(); torch.cuda.synchronize()
    ok = torch.equal(outs[-1], ref)
    graph = timeit(g.replay)
    print("resnet%d e=%d: eager %.3f ms, graph replay %.3f ms, same result: %s" % (version, e, eager, graph, ok))
