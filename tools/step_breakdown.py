"""Where a training step's wall time goes on the MAIN stream: HIP events around forward / loss / backward of the online
loop body (two-stream backward with deferred join, as train_online._train runs it).  usage: python tools/step_breakdown.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))
from dataloaders.synthetic import make_frame  # noqa: E402
from layers.osvos_layers import class_balanced_cross_entropy_loss  # noqa: E402
from networks.osvos_vgg import OSVOS_VGG  # noqa: E402
import parallel  # noqa: E402
from util.network_provider import provider_mapping  # noqa: E402,F401

torch.manual_seed(0)
net = OSVOS_VGG(pretrained=0).cuda()
if os.environ.get("STEP_INIT", "kaiming") == "kaiming":  # "ref": the reference's N(0, 0.001) init (what bench.py loads)
    for n, p in net.named_parameters():
        if ("stages" in n or "side_prep" in n) and "weight" in n:
            torch.nn.init.kaiming_normal_(p)
net.accumulate_grads_in_place = True
net.compute_side_outputs = False
net.defer_wgrad_join = True
flat = parallel.FlatGrads(net.parameters())
img, gt = make_frame(480, 854)
x, y = img.unsqueeze(0).cuda(), gt.unsqueeze(0).cuda()
ev = lambda: torch.cuda.Event(enable_timing=True)
N = 60
if os.environ.get("STEP_EVENTS", "1") == "0":  # plain wall clock of the same loop, no events
    import time
    def body(it):
        out = net.forward(x)
        loss = class_balanced_cross_entropy_loss(out[-1], y, size_average=False) / 5
        loss.backward()
        if it % 5 == 4:
            net.join_gradients()
            flat.zero()
    for it in range(10):
        body(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(10, 10 + 5 * N):
        body(it)
    torch.cuda.synchronize()
    print("wall clock, no events: %.3f ms/step" % ((time.perf_counter() - t0) / (5 * N) * 1e3))
    sys.exit(0)
marks = []
for it in range(N + 10):
    e = [ev() for _ in range(5)]
    e[0].record()
    out = net.forward(x)
    e[1].record()
    loss = class_balanced_cross_entropy_loss(out[-1], y, size_average=False) / 5
    e[2].record()
    loss.backward()
    e[3].record()
    if it % 5 == 4:
        net.join_gradients()
        flat.zero()
    e[4].record()
    marks.append(e)
torch.cuda.synchronize()  # one sync at the end: the host runs ahead, the events time the main stream as it really executes
acc = {"forward": 0.0, "loss": 0.0, "backward (main-stream span)": 0.0, "join + zero": 0.0, "gap to next step": 0.0}
for it in range(10, N + 10):
    e = marks[it]
    acc["forward"] += e[0].elapsed_time(e[1])
    acc["loss"] += e[1].elapsed_time(e[2])
    acc["backward (main-stream span)"] += e[2].elapsed_time(e[3])
    acc["join + zero"] += e[3].elapsed_time(e[4])
    if it + 1 < N + 10:
        acc["gap to next step"] += e[4].elapsed_time(marks[it + 1][0])
for k, v in acc.items():
    print("%-28s %.3f ms/step" % (k, v / N))
print("%-28s %.3f ms/step" % ("sum", sum(acc.values()) / N))
