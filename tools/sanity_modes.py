"""GPU sanity/timing of the two modes bench.py does not time: forward-only inference at 854x480 and an
offline-style step (N=4, five class-balanced losses, src/train_offline.py:77-110).  usage: python tools/sanity_modes.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.getcwd(), "fosvos_amd"))
from networks.osvos_vgg import OSVOS_VGG
from layers.osvos_layers import class_balanced_cross_entropy_loss
from dataloaders.synthetic import make_frame
torch.manual_seed(0)
net = OSVOS_VGG(pretrained=0).cuda()
# kaiming-ish init so activations are alive
for n, p in net.named_parameters():
    if 'stages' in n and 'weight' in n:
        torch.nn.init.kaiming_normal_(p)
# inference: forward only, 854x480
img, gt = make_frame(480, 854)
x = img.unsqueeze(0).cuda(); y = gt.unsqueeze(0).cuda()
with torch.no_grad():
    for _ in range(5): out = net.forward(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): out = net.forward(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print("inference (forward, 5 logit maps) ms/frame: %.3f  -> %.0f frames/s" % (dt * 1e3, 1 / dt))
# offline-style step at N=4: 5 deep-supervised losses
xb = torch.stack([make_frame(480, 854, index=i)[0] for i in range(4)]).cuda()
yb = torch.stack([make_frame(480, 854, index=i)[1] for i in range(4)]).cuda()
for it in range(3):
    outs = net.forward(xb)
    losses = [class_balanced_cross_entropy_loss(o, yb, size_average=False) for o in outs]
    loss = 0.5 * sum(losses[:-1]) + losses[-1]
    (loss / 10).backward()
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(5):
    outs = net.forward(xb)
    losses = [class_balanced_cross_entropy_loss(o, yb, size_average=False) for o in outs]
    loss = 0.5 * sum(losses[:-1]) + losses[-1]
    (loss / 10).backward()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("offline step N=4 (5 losses, fwd+bwd) ms: %.2f -> %.0f frames/s; loss %.4g; grad finite: %s" % (
    dt * 1e3, 4 / dt, float(loss.detach()), all(torch.isfinite(p.grad).all().item() for p in net.parameters() if p.grad is not None)))
print("max memory GB: %.2f" % (torch.cuda.max_memory_allocated() / 1e9))
