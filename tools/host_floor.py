#!/usr/bin/env python3
"""How long does the HOST need to issue one training step (no waiting for the GPU)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))
import torch
import fosvos_amd  # noqa
from dataloaders.synthetic import make_frame
from layers.osvos_layers import class_balanced_cross_entropy_loss as cbce
from networks.osvos_vgg import OSVOS_VGG
from util.network_provider import VGGOnlineProvider
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 854)
dev = "cuda:0"
torch.manual_seed(0)
net = OSVOS_VGG(pretrained=0).to(dev)
net.accumulate_grads_in_place = True
prov = VGGOnlineProvider.__new__(VGGOnlineProvider); prov.network = net
opt = prov.get_optimizer()
img, gt = make_frame(H, W); x, y = img.unsqueeze(0).to(dev), gt.unsqueeze(0).to(dev)
def step(i):
    loss = cbce(net(x)[-1], y, size_average=False)
    (loss / 5).backward()
    if (i + 1) % 5 == 0:
        opt.step(); opt.zero_grad()
for i in range(10): step(i)
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for i in range(n): step(i)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"{H}x{W}: host issue {t_issue/n*1e3:.3f} ms/step, end-to-end {t_all/n*1e3:.3f} ms/step")
if os.environ.get("PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(20): step(i)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
