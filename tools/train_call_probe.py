import os, sys, time
sys.path.insert(0, "fosvos_amd"); sys.path.insert(0, ".")
import torch
import train_online, parallel
from dataloaders.synthetic import make_frame
from util.network_provider import VGGOnlineProvider
from networks.osvos_vgg import OSVOS_VGG
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = OSVOS_VGG(pretrained=0)
with torch.no_grad():
    for name, p in net.named_parameters():
        if name.startswith("upscale"): continue
        if p.dim() == 4:
            fan_in = p.shape[1] * p.shape[2] * p.shape[3]
            p.normal_(0, (2.0 / fan_in) ** 0.5 if name.startswith("stages") else (1.0 / fan_in) ** 0.5)
        else: p.normal_(0, 0.1)
prov = VGGOnlineProvider.__new__(VGGOnlineProvider); prov.network = net.to(dev); prov.name = "vgg16"
opt = prov.get_optimizer()
img, gt = make_frame(480, 854, seed=1234, index=0)
batch = [{"image": img.unsqueeze(0).to(dev), "gt": gt.unsqueeze(0).to(dev)}]
class W:
    def add_scalar(self, *a, **k): pass
def run(n):
    torch.cuda.synchronize(); t = time.perf_counter()
    train_online._train(prov, batch, opt, W(), "t", 0, n, 5, 10 ** 9)
    torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
mode = sys.argv[1] if len(sys.argv) > 1 else "series"
if mode == "series":
    run(10)
    for n in (5, 10, 20, 40, 5, 10, 20, 40):
        print(n, "steps: %.2f ms" % run(n))
else:  # what a short bench run sees: a cold device, W warm-up steps, K timed steps
    if mode == "preheat":
        a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
        t = time.perf_counter()
        while time.perf_counter() - t < 0.05:
            for _ in range(10):
                a @ a
            torch.cuda.synchronize()
    w = run(5)
    print(mode, "warm-up 5 steps: %.2f ms; timed 20 steps: %.2f ms" % (w, run(20)))
    print(mode, "again 20 steps: %.2f ms" % run(20))
    time.sleep(0.3)
    print(mode, "after 300 ms of idle, 20 steps: %.2f ms" % run(20))
    print(mode, "again 20 steps: %.2f ms" % run(20))
    for _ in range(3):
        run(20)
    time.sleep(0.02)
    print(mode, "after 20 ms of idle, 20 steps: %.2f ms" % run(20))
    sys.exit(0)
named = list(prov.network.named_parameters())
torch.cuda.synchronize(); t = time.perf_counter()
flat = parallel.FlatGrads([p for _, p in named], names=[n for n, _ in named]); torch.cuda.synchronize()
print("FlatGrads: %.3f ms" % ((time.perf_counter() - t) * 1e3))
t = time.perf_counter(); r = torch.empty((64, 5), dtype=torch.float32).pin_memory(); print("pinned ring: %.3f ms" % ((time.perf_counter() - t) * 1e3))
t = time.perf_counter(); torch.cuda.synchronize(); print("sync: %.3f ms" % ((time.perf_counter() - t) * 1e3))
