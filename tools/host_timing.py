#!/usr/bin/env python3
"""Where the HOST time of the online loop goes (GPU box): wraps every C-ABI entry point of libfosvos_hip.so with a
perf_counter pair, runs the bench's training loop, prints per-call host time and the loop's enqueue time vs the
device's completion time.   usage: host_timing.py [steps]"""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import fosvos_hip  # noqa: E402

acc = collections.defaultdict(lambda: [0, 0.0])


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    cdll = fosvos_hip.lib()
    import ctypes
    # every exported fosvos_* symbol the binding has touched so far is an attribute of the CDLL
    for name in list(vars(cdll).keys()):
        fn = getattr(cdll, name)
        if not name.startswith("fosvos_") or not isinstance(fn, ctypes._CFuncPtr):
            continue

        def make(fn, name):
            def timed(*a):
                t = time.perf_counter()
                r = fn(*a)
                d = time.perf_counter() - t
                e = acc[name]
                e[0] += 1
                e[1] += d
                return r
            return timed
        setattr(cdll, name, make(fn, name))

    import train_online
    from dataloaders.synthetic import make_frame
    dev = torch.device("cuda:0")
    if True:
        from util.network_provider import VGGOnlineProvider
        from networks.osvos_vgg import OSVOS_VGG
        torch.manual_seed(0)
        prov = VGGOnlineProvider.__new__(VGGOnlineProvider)
        net = OSVOS_VGG(pretrained=0)
        with torch.no_grad():  # bench.py's variance-preserving init (O(1) activations: realistic device clocks)
            for name, p in net.named_parameters():
                if name.startswith("upscale"):
                    continue
                if p.dim() == 4:
                    fan_in = p.shape[1] * p.shape[2] * p.shape[3]
                    p.normal_(0, (2.0 / fan_in) ** 0.5 if name.startswith("stages") else (1.0 / fan_in) ** 0.5)
                else:
                    p.normal_(0, 0.1)
        prov.network = net.to(dev)
        prov.name = "vgg16"
        opt = prov.get_optimizer()
    img, gt = make_frame(480, 854, seed=1234, index=0)
    batch = [{"image": img.unsqueeze(0).to(dev), "gt": gt.unsqueeze(0).to(dev)}]

    class W:
        def add_scalar(self, *a, **k):
            pass

    train_online._train(prov, batch, opt, W(), "t", 0, 10, 5, 10 ** 9)
    torch.cuda.synchronize()
    acc.clear()
    t0 = time.perf_counter()
    ret = train_online._train(prov, batch, opt, W(), "t", 0, steps, 5, 10 ** 9)
    total = time.perf_counter() - t0
    print(f"{steps} steps: device done after {total * 1e3:.1f} ms, host loop enqueued in {ret['seconds_host_enqueue'] * 1e3:.1f} ms")
    tot_c = 0.0
    for name, (n, sec) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"  {name:38s} {n:5d} calls  {sec * 1e3:8.2f} ms  {sec / n * 1e6:8.1f} us/call")
        tot_c += sec
    print(f"  C-ABI calls total {tot_c * 1e3:.1f} ms; Python + torch around them {(ret['seconds_host_enqueue'] - tot_c) * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
