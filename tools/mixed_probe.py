"""Lab tool: the mixed-size online loop (bench.py's mixed_scales loader) - wall time and host enqueue time of 100 steps.
   python tools/mixed_probe.py      (FOSVOS_PASS_STREAMS=0/1 etc. from the environment)"""
import random, sys, time
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
rng = random.Random(1234)
scales = [rng.choice((1.0, 0.8, 0.5)) for _ in range(20)]
mixed = []
for i, sc in enumerate(scales):
    im, g = make_frame(int(480 * sc), int(854 * sc), seed=1234, index=i)
    mixed.append({"image": im.unsqueeze(0).to(dev), "gt": g.unsqueeze(0).to(dev)})
class W:
    def add_scalar(self, *a, **k): pass
def run(epochs):
    torch.cuda.synchronize(); t = time.perf_counter()
    r = train_online._train(prov, mixed, opt, W(), "t", 0, epochs, 5, 10 ** 9)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3, r["seconds_host_enqueue"] * 1e3, r["iterations"]
import os
if len(sys.argv) > 1 and sys.argv[1] == "g1first":
    full = [m for m in mixed if m["image"].shape[2] == 480][:1]
    os.environ["FOSVOS_MICROBATCH_GROUP"] = "1"
    torch.cuda.synchronize(); t = time.perf_counter()
    train_online._train(prov, full, opt, W(), "t", 0, 40, 5, 10 ** 9)
    torch.cuda.synchronize(); print("single-frame passes, 40 steps: %.1f ms" % ((time.perf_counter() - t) * 1e3))
    del os.environ["FOSVOS_MICROBATCH_GROUP"]
run(2)
for _ in range(3):
    ms, host, it = run(5)
    print(f"{it} steps: {ms:.1f} ms wall ({it / ms * 1e3:.0f} frames/s), host enqueue {host:.1f} ms")
print("window groups of the first cycles:", [[s for s in scales[i:i + 5]] for i in (0, 5, 10, 15)])
