#!/usr/bin/env python3
"""Forward-only timing at 480x854 (GPU box): pipelined (no sync between frames) and the reference's eval_speeds protocol
(sync around every forward), on a fresh net and on a net that has just been fine-tuned."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))
import fosvos_amd  # noqa
import train_online
from dataloaders.synthetic import make_frame
from networks.osvos_vgg import OSVOS_VGG
from util.network_provider import VGGOnlineProvider

def protocol(net, x, n=30):
    ts = []
    with torch.no_grad():
        for i in range(n + 3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            net.forward(x)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts = sorted(ts[3:])
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3

def pipelined(net, x, n=50):
    with torch.no_grad():
        for _ in range(5): net.forward(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): net.forward(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

torch.manual_seed(0)
net = OSVOS_VGG(pretrained=0).cuda()
for n_, p in net.named_parameters():
    if 'stages' in n_ and 'weight' in n_: torch.nn.init.kaiming_normal_(p)
img, gt = make_frame(480, 854)
x = img.unsqueeze(0).cuda()
print("fresh net: pipelined %.3f ms/frame; protocol median %.3f min %.3f ms" % ((pipelined(net, x),) + protocol(net, x)))
prov = VGGOnlineProvider.__new__(VGGOnlineProvider); prov.network = net; prov.name = "vgg16"
opt = prov.get_optimizer()
class W:
    def add_scalar(self, *a, **k): pass
batch = [{"image": x, "gt": gt.unsqueeze(0).cuda()}]
train_online._train(prov, batch, opt, W(), "lab", 0, 20, 5, 10 ** 9)
print("after _train: pipelined %.3f ms/frame; protocol median %.3f min %.3f ms" % ((pipelined(net, x),) + protocol(net, x)))
print("max memory GB: %.2f" % (torch.cuda.max_memory_allocated() / 1e9))
