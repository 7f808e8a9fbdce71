"""Lab tool: where the fixed cost of one train_online._train call goes (host side): entry -> first net.forward, last
optimizer step -> return, and the device-side wall time of 5 / 20 steps.   python tools/train_startup_probe.py"""
import sys, time
sys.path.insert(0, "fosvos_amd"); sys.path.insert(0, ".")
import torch
import train_online
from dataloaders.synthetic import make_frame
from util.network_provider import VGGOnlineProvider
from networks.osvos_vgg import OSVOS_VGG
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = OSVOS_VGG(pretrained=0)
prov = VGGOnlineProvider.__new__(VGGOnlineProvider); prov.network = net.to(dev); prov.name = "vgg16"
opt = prov.get_optimizer()
img, gt = make_frame(480, 854, seed=1234, index=0)
batch = [{"image": img.unsqueeze(0).to(dev), "gt": gt.unsqueeze(0).to(dev)}]
class W:
    def add_scalar(self, *a, **k): pass
marks = {}
orig_fwd = net.forward
def fwd(x):
    marks.setdefault("first_forward", time.perf_counter())
    marks["last_forward"] = time.perf_counter()
    return orig_fwd(x)
net.forward = fwd
orig_step = opt.step
def step(*a, **k):
    r = orig_step(*a, **k)
    marks["last_step"] = time.perf_counter()
    return r
opt.step = step
def run(n):
    marks.clear()
    torch.cuda.synchronize(); t = time.perf_counter()
    train_online._train(prov, batch, opt, W(), "t", 0, n, 5, 10 ** 9)
    t_ret = time.perf_counter()
    torch.cuda.synchronize(); t_end = time.perf_counter()
    return dict(total=(t_end - t) * 1e3, to_first_forward=(marks["first_forward"] - t) * 1e3,
                host_return=(t_ret - t) * 1e3, after_last_step=(t_ret - marks["last_step"]) * 1e3,
                device_tail_after_return=(t_end - t_ret) * 1e3)
run(10); run(10)
for n in (5, 20, 20, 40):
    print(n, {k: round(v, 3) for k, v in run(n).items()})
if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(20):
        train_online._train(prov, batch, opt, W(), "t", 0, 5, 5, 10 ** 9)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
