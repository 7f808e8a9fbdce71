#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

The reference (read-only at /root/reference) is imported with three non-arithmetic stubs
(colorlog, config.mypath, torchvision.models symbols that are never called) exactly as
SURVEY.md §8(c) describes; every number written here comes out of the reference's own
``OSVOS_VGG``, ``class_balanced_cross_entropy_loss``, ``center_crop``, ``upsample_filt``,
``interp_surgery``, ``_make_layers_osvos`` and ``VGG{Online,Offline}Provider.get_optimizer``.
The fixtures are data only (inputs, seeds and expected outputs).  The reference never travels
to the GPU box; the fixtures and this script do.

Usage:  python oracle/make_golden.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import logging
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
REF_SRC = "/root/reference/src"


def _install_stubs() -> None:
    cl = types.ModuleType("colorlog")
    cl.StreamHandler = logging.StreamHandler
    cl.ColoredFormatter = lambda *a, **k: logging.Formatter()
    cl.getLogger = logging.getLogger
    sys.modules["colorlog"] = cl

    sys.path.insert(0, REF_SRC)
    import config  # the reference's own (empty) package

    mp = types.ModuleType("config.mypath")

    class Path:  # the real file is git-ignored and absent from the reference
        @staticmethod
        def models_dir():
            return "/nonexistent"

    mp.Path = Path
    sys.modules["config.mypath"] = mp
    config.mypath = mp

    def _never(*a, **k):
        raise RuntimeError("torchvision is not available offline")

    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvr = types.ModuleType("torchvision.models.resnet")
    for n in ("vgg16", "resnet18", "resnet34", "resnet50", "resnet101", "resnet152"):
        setattr(tvm, n, _never)
    for n in ("BasicBlock", "Bottleneck", "ResNet"):
        setattr(tvr, n, type(n, (torch.nn.Module,), {}))
    tv.models = tvm
    tvm.resnet = tvr
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm
    sys.modules["torchvision.models.resnet"] = tvr


def digest(t: torch.Tensor, n_samples: int = 64):
    """Size-independent summary of a big tensor: float64 moments + strided samples."""
    f = t.detach().reshape(-1).to(torch.float64)
    n = f.numel()
    idx = torch.linspace(0, n - 1, steps=min(n, n_samples)).round().long()
    return (np.array([f.sum().item(), f.abs().sum().item(), (f * f).sum().item()], dtype=np.float64),
            idx.numpy().astype(np.int64), t.detach().reshape(-1)[idx].numpy().copy())


def full_or_dense(store, prefix, name, d):
    """Round 3: a 64-sample digest estimates a tensor's relative L2 error only to a few tens of percent, which is what the
    loop tests' single-tensor slack used to pay for.  Tensors of at most 64 k elements are stored whole (fp32: the deltas are
    differences of fp32 weights), the larger ones with 4096 strided samples."""
    if d.numel() <= 65536:
        store[f"{prefix}_fulldelta_{name}"] = d.to(torch.float32).numpy().copy()
    else:
        _, idx, smp = digest(d, n_samples=4096)
        store[f"{prefix}_dense_{name}_i"] = idx
        store[f"{prefix}_dense_{name}_s"] = smp.astype(np.float32)


def trajectory_fixture(OSVOS_VGG, RL, RNP, O, TRAJ=None, compact=False):
    """compact (the 480x854 fixture): logit maps as fp16 + their fp32 absolute maximum, full deltas as fp32."""
    TRAJ = O.TRAJ if TRAJ is None else TRAJ
    sd0, frames, (xh, gh) = O.trajectory_inputs(TRAJ)
    net = OSVOS_VGG(pretrained=0)
    net.load_state_dict(sd0)
    cls = RNP.VGGOnlineProvider
    prov = cls.__new__(cls)
    prov.network = net
    opt = cls.get_optimizer(prov, learning_rate=TRAJ["lr"])
    with torch.no_grad():
        start = net.forward(xh)[-1]
    trace, counter = [], 0
    for it in range(TRAJ["iters"]):  # src/train_online.py:70-101
        x, gt = frames[it % len(frames)]
        outs = net.forward(x)
        loss = RL.class_balanced_cross_entropy_loss(outs[-1], gt, size_average=False)
        trace.append(loss.item())
        loss = loss / TRAJ["avg"]
        loss.backward()
        counter += 1
        if counter % TRAJ["avg"] == 0:
            opt.step()
            opt.zero_grad()
            counter = 0
    with torch.no_grad():  # the test pass of train_and_test (src/train_online.py:36-47) on a frame the loop never saw
        held = net.forward(xh)[-1]
        seen = net.forward(frames[0][0])[-1]
    out = {k: np.float64(v) if isinstance(v, float) else np.int64(v) for k, v in TRAJ.items()}
    out["loss"] = np.array(trace, dtype=np.float64)
    if compact:
        out["heldout_logits_f16"] = held[0, 0].numpy().astype(np.float16)
        out["heldout_logits_absmax"] = np.float32(held.abs().max().item())
        out["heldout_start_mask_bits"] = np.packbits((start[0, 0] >= 0).numpy())
        out["train_mask_bits"] = np.packbits((seen[0, 0] >= 0).numpy())
    else:
        out["heldout_logits_start"] = start[0, 0].numpy().copy()
        out["heldout_logits"] = held[0, 0].numpy().copy()
        out["train_logits"] = seen[0, 0].numpy().copy()
    out["heldout_mask_bits"] = np.packbits((held[0, 0] >= 0).numpy())
    out["heldout_gt_bits"] = np.packbits((gh[0, 0] > 0.5).numpy())
    full, digested = [], []
    for name, p in net.named_parameters():
        d = p.detach().double() - sd0[name].double()
        if d.numel() <= 65536:
            out[f"delta_{name}"] = d.numpy().astype(np.float32 if compact else np.float64)
            full.append(name)
        else:
            m, idx, smp = digest(d, n_samples=4096)
            out[f"delta_{name}_m"], out[f"delta_{name}_i"], out[f"delta_{name}_s"] = m, idx, smp
            digested.append(name)
    out["full_tensors"] = np.array(full)
    out["digest_tensors"] = np.array(digested)
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(HERE), "tests", "golden"))
    ap.add_argument("--skip-e2e", action="store_true")
    ap.add_argument("--only", default=None,
                    help="write only this fixture (kat, stacks, net, loops, e2e, trajectory, bwd_full, trajectory_full)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    logging.disable(logging.CRITICAL)
    _install_stubs()
    torch.manual_seed(0)
    torch.set_num_threads(max(1, os.cpu_count() or 1))

    from layers import osvos_layers as RL          # reference
    from networks.osvos_vgg import OSVOS_VGG       # reference
    from util import network_provider as RNP       # reference
    from oracle import osvos_ref as O              # ours: only for seeds / synthetic inputs

    # ------------------------------------------------------------------ 1. known-answer tests
    kat = {}
    for k in (3, 4, 5, 8, 16, 32):
        kat[f"filt_{k}"] = RL.upsample_filt(k)
    for c, k in ((16, 4), (1, 8), (3, 16)):
        lay = torch.nn.ConvTranspose2d(c, c, k, stride=k // 2, bias=False)
        lay.weight.data.zero_()
        kat[f"surgery_{c}_{k}"] = RL.interp_surgery(lay).numpy().copy()
    crop_src = torch.arange(2 * 3 * 11 * 14, dtype=torch.float32).view(2, 3, 11, 14)
    kat["crop_src"] = crop_src.numpy()
    crop_cases = [(11, 14), (10, 13), (9, 12), (8, 9), (5, 14), (11, 3), (4, 5)]
    kat["crop_cases"] = np.array(crop_cases, dtype=np.int64)
    for h, w in crop_cases:
        kat[f"crop_{h}_{w}"] = RL.center_crop(crop_src, h, w).numpy().copy()

    def loss_case(tag, x, y):
        x = x.clone().requires_grad_(True)
        l_sum = RL.class_balanced_cross_entropy_loss(x, y, size_average=False)
        (g_sum,) = torch.autograd.grad(l_sum, x)
        l_avg = RL.class_balanced_cross_entropy_loss(x, y, size_average=True)
        kat[f"loss_{tag}_x"] = x.detach().numpy().copy()
        kat[f"loss_{tag}_y"] = y.numpy().copy()
        kat[f"loss_{tag}_sum"] = np.float32(l_sum.item())
        kat[f"loss_{tag}_avg"] = np.float32(l_avg.item())
        kat[f"loss_{tag}_grad"] = g_sum.numpy().copy()

    y12 = torch.tensor([0, 0, 1, 0, 0, 1, 1, 0, 0, 0, .5, .49]).view(1, 1, 3, 4)
    loss_case("kat12", torch.linspace(-3, 3, 12).view(1, 1, 3, 4), y12)
    loss_case("allneg", torch.linspace(-3, 3, 12).view(1, 1, 3, 4), torch.zeros(1, 1, 3, 4))
    loss_case("allpos", torch.linspace(-3, 3, 12).view(1, 1, 3, 4), torch.ones(1, 1, 3, 4))
    loss_case("extreme", torch.tensor([-100., 100., 0.]).view(1, 1, 1, 3), torch.tensor([1., 0., 1.]).view(1, 1, 1, 3))
    g = torch.Generator().manual_seed(7)
    loss_case("rand", torch.randn(2, 1, 13, 17, generator=g) * 4,
              (torch.rand(2, 1, 13, 17, generator=g) > 0.8).float())
    loss_case("soft", torch.randn(1, 1, 9, 31, generator=g) * 30, torch.rand(1, 1, 9, 31, generator=g))
    if args.only in (None, "kat"):
        np.savez_compressed(os.path.join(args.out, "kat.npz"), **kat)

    # ------------------------------------------------------------------ 2. per-op layer stacks
    # built by the reference's own _make_layers_osvos at ragged sizes (ceil-mode pool edges)
    stacks = {}
    g = torch.Generator().manual_seed(11)
    cases = [("a", ["M", 8, 8], 4, (1, 61, 107)), ("b", ["M", 16], 8, (2, 30, 54)),
             ("c", [8, 8], 3, (1, 48, 86)), ("d", ["M", 8, 8, 8], 8, (1, 7, 5)),
             ("e", ["M", 8], 8, (1, 1, 1))]
    for tag, cfg, cin, (n, h, w) in cases:
        seq = OSVOS_VGG._make_layers_osvos(cfg, cin)
        for p in seq.parameters():
            p.data = torch.randn(p.shape, generator=g) * (0.3 if p.dim() == 4 else 0.1)
        x = torch.randn(n, cin, h, w, generator=g).requires_grad_(True)
        y = seq(x)
        gy = torch.randn(y.shape, generator=g)
        grads = torch.autograd.grad(y, [x] + list(seq.parameters()), gy)
        stacks[f"{tag}_cfg"] = np.array([-1 if v == "M" else v for v in cfg], dtype=np.int64)
        stacks[f"{tag}_cin"] = np.int64(cin)
        stacks[f"{tag}_x"] = x.detach().numpy().copy()
        stacks[f"{tag}_y"] = y.detach().numpy().copy()
        stacks[f"{tag}_gy"] = gy.numpy().copy()
        stacks[f"{tag}_gx"] = grads[0].numpy().copy()
        for i, (p, gp) in enumerate(zip(seq.parameters(), grads[1:])):
            stacks[f"{tag}_p{i}"] = p.detach().numpy().copy()
            stacks[f"{tag}_gp{i}"] = gp.numpy().copy()
    if args.only in (None, "stacks"):
        np.savez_compressed(os.path.join(args.out, "stacks.npz"), **stacks)

    # ------------------------------------------------------------------ 3. full network fwd + bwd
    def ref_net(seed, scheme="kaiming"):
        net = OSVOS_VGG(pretrained=0)
        keys = list(net.state_dict().keys())
        sd = O.make_state_dict(seed, scheme)
        assert keys == list(sd.keys()), "state_dict key order differs from the reference"
        for k_, v in net.state_dict().items():
            assert tuple(v.shape) == tuple(sd[k_].shape), k_
        net.load_state_dict(sd)
        return net, sd

    # the reference's own initialisation of the deconvs must equal what the oracle seeds
    net0 = OSVOS_VGG(pretrained=0)
    netfix = {"keys": np.array(list(net0.state_dict().keys())),
              "shapes": np.array([str(tuple(v.shape)) for v in net0.state_dict().values()])}
    for i in range(4):
        netfix[f"init_upscale_{i}_diag"] = net0.upscale[i].weight.data[3, 3].numpy().copy()
        netfix[f"init_upscale_{i}_offdiag_absmax"] = np.float32(net0.upscale[i].weight.data[3, 5].abs().max().item())
        netfix[f"init_upscale__{i}"] = net0.upscale_[i].weight.data[0, 0].numpy().copy()
    netfix["init_conv_std"] = np.float32(net0.stages[2][1].weight.data.std().item())
    netfix["init_bias_absmax"] = np.float32(net0.stages[2][1].bias.data.abs().max().item())

    for tag, seed, (n, h, w) in (("s", 3, (1, 48, 86)), ("r", 4, (2, 61, 107))):
        net, _ = ref_net(seed)
        x, gt = O.synthetic_frame(n, h, w, seed=100 + seed)
        outs = net.forward(x)
        losses = [RL.class_balanced_cross_entropy_loss(o, gt, size_average=False) for o in outs]
        netfix[f"{tag}_seed"] = np.int64(seed)
        netfix[f"{tag}_shape"] = np.array([n, h, w], dtype=np.int64)
        netfix[f"{tag}_frame_seed"] = np.int64(100 + seed)
        for i, o in enumerate(outs):
            netfix[f"{tag}_out{i}"] = o.detach().numpy().copy()
            netfix[f"{tag}_loss{i}"] = np.float64(losses[i].item())
        # online objective: fused loss only (src/train_online.py:81)
        net.zero_grad()
        losses[-1].backward(retain_graph=True)
        for name, p in net.named_parameters():
            if p.grad is None:
                continue
            m, idx, smp = digest(p.grad)
            netfix[f"{tag}_on_g_{name}_m"] = m
            netfix[f"{tag}_on_g_{name}_i"] = idx
            netfix[f"{tag}_on_g_{name}_s"] = smp
        netfix[f"{tag}_on_nograd"] = np.array([nm for nm, p in net.named_parameters() if p.grad is None])
        # offline objective at epoch 60/240 (src/train_offline.py:88)
        net.zero_grad()
        ((1 - 60 / 240) * sum(losses[:-1]) + losses[-1]).backward()
        for name, p in net.named_parameters():
            m, idx, smp = digest(p.grad)
            netfix[f"{tag}_off_g_{name}_m"] = m
            netfix[f"{tag}_off_g_{name}_i"] = idx
            netfix[f"{tag}_off_g_{name}_s"] = smp
    if args.only in (None, "net"):
        np.savez_compressed(os.path.join(args.out, "net.npz"), **netfix)

    # ------------------------------------------------------------------ 4. optimizer recipe + loops
    loops = {}

    def provider(cls, net):
        prov = cls.__new__(cls)
        prov.network = net
        return prov

    for mode, cls in (("online", RNP.VGGOnlineProvider), ("offline", RNP.VGGOfflineProvider)):
        net, _ = ref_net(5)
        opt = cls.get_optimizer(provider(cls, net))
        names = {id(p): n for n, p in net.named_parameters()}
        rows = []
        for gi, grp in enumerate(opt.param_groups):
            for p in grp["params"]:
                rows.append(f"{gi}|{names[id(p)]}|{grp['lr']!r}|{grp['weight_decay']!r}|{grp['momentum']!r}")
        loops[f"groups_{mode}"] = np.array(rows)

    # online: 10 iterations, step every 5 (src/train_online.py:70-101), two alternating frame sizes
    for tag, lr in (("lr1e-8", 1e-8), ("lr1e-9", 1e-9)):
        net, sd0 = ref_net(6)
        cls = RNP.VGGOnlineProvider
        opt = cls.get_optimizer(provider(cls, net), learning_rate=lr)
        frames = [O.synthetic_frame(1, 48, 86, seed=21), O.synthetic_frame(1, 40, 70, seed=22)]
        trace = []
        counter = 0
        for it in range(10):
            x, gt = frames[it % 2]
            outs = net.forward(x)
            loss = RL.class_balanced_cross_entropy_loss(outs[-1], gt, size_average=False)
            trace.append(loss.item())
            loss = loss / 5
            loss.backward()
            counter += 1
            if counter % 5 == 0:
                opt.step()
                opt.zero_grad()
                counter = 0
        loops[f"online_{tag}_loss"] = np.array(trace, dtype=np.float64)
        for name, p in net.named_parameters():
            d = (p.detach().double() - sd0[name].double())
            m, idx, smp = digest(d)
            loops[f"online_{tag}_delta_{name}_m"] = m
            loops[f"online_{tag}_delta_{name}_i"] = idx
            loops[f"online_{tag}_delta_{name}_s"] = smp
            full_or_dense(loops, f"online_{tag}", name, d)
        loops[f"online_{tag}_fuse_weight"] = net.fuse.weight.detach().numpy().copy()
        loops[f"online_{tag}_stage00_bias"] = net.stages[0][0].bias.detach().numpy().copy()

    # offline: 4 iterations, step every 2, epoch 60 of 240 (src/train_offline.py:77-110)
    net, sd0 = ref_net(8)
    cls = RNP.VGGOfflineProvider
    opt = cls.get_optimizer(provider(cls, net), learning_rate=1e-6)
    x, gt = O.synthetic_frame(2, 33, 47, seed=23)
    trace = []
    counter = 0
    for it in range(4):
        outs = net.forward(x)
        ls = [RL.class_balanced_cross_entropy_loss(o, gt, size_average=False) for o in outs]
        trace.append([l.item() for l in ls])
        loss = (1 - 60 / 240) * sum(ls[:-1]) + ls[-1]
        loss = loss / 2
        loss.backward()
        counter += 1
        if counter % 2 == 0:
            opt.step()
            opt.zero_grad()
            counter = 0
    loops["offline_loss"] = np.array(trace, dtype=np.float64)
    for name, p in net.named_parameters():
        d = (p.detach().double() - sd0[name].double())
        m, idx, smp = digest(d)
        loops[f"offline_delta_{name}_m"] = m
        loops[f"offline_delta_{name}_i"] = idx
        loops[f"offline_delta_{name}_s"] = smp
        full_or_dense(loops, "offline", name, d)
    if args.only in (None, "loops"):
        np.savez_compressed(os.path.join(args.out, "loops.npz"), **loops)

    # ------------------------------------------------------------------ 5. one 854x480 frame end to end
    if not args.skip_e2e and args.only in (None, "e2e"):
        net, _ = ref_net(9)
        x, gt = O.synthetic_frame(1, 480, 854, seed=1234)
        with torch.no_grad():
            outs = net.forward(x)
        fused = outs[-1][0, 0]
        e2e = {"seed": np.int64(9), "frame_seed": np.int64(1234),
               "logits_f16": fused.numpy().astype(np.float16),
               "logits_absmax": np.float32(fused.abs().max().item()),
               "mask_bits": np.packbits((fused >= 0).numpy()),
               "side_absmax": np.array([o.abs().max().item() for o in outs[:4]], dtype=np.float32),
               "loss_fused_sum": np.float64(RL.class_balanced_cross_entropy_loss(outs[-1], gt, size_average=False).item())}
        for i in range(4):
            m, idx, smp = digest(outs[i])
            e2e[f"side{i}_m"] = m
            e2e[f"side{i}_i"] = idx
            e2e[f"side{i}_s"] = smp
        np.savez_compressed(os.path.join(args.out, "e2e_480x854.npz"), **e2e)
    # ------------------------------------------------------------------ 6. fine-tune TRAJECTORY (train, then test)
    # The north-star parity: masks of reference-fine-tuned weights (src/train_online.py:23-50: _train, then test).  The
    # reference's own OSVOS_VGG + VGGOnlineProvider.get_optimizer fine-tune for TRAJ_ITERS iterations (step every 5) on
    # the augmentations of one annotated frame (the frame and its horizontal flip: custom_transforms.RandomHorizontalFlip
    # with the draw fixed), from a seeded parent whose head is scaled down so that the logits start O(1) and the run is
    # smooth (loss 3076 -> ~170, the held-out mask moves from IoU 0.00 to 0.88 against the annotation) - then the
    # fused logits of a HELD-OUT frame (the object displaced) are stored whole, with full deltas of every tensor of at most
    # 64 k elements (digests for the larger ones).
    if args.only in (None, "trajectory"):
        traj = trajectory_fixture(OSVOS_VGG, RL, RNP, O)
        np.savez_compressed(os.path.join(args.out, "trajectory.npz"), **traj)
    # ------------------------------------------------------------------ 7. backward pass at the BASELINE frame size
    # ONE reference forward + class-balanced loss + backward of the online objective (src/train_online.py:79-93,
    # src/networks/osvos_vgg.py:61-83) on the 1x3x480x854 frame of section 5, same net: every gradient - tensors of at
    # most 64 k elements whole, the larger ones as 4096 strided samples - plus float64 moments (sum, sum |g|, sum g^2) of
    # ALL elements of every tensor.
    if not args.skip_e2e and args.only in (None, "bwd_full"):
        net, _ = ref_net(9)
        x, gt = O.synthetic_frame(1, 480, 854, seed=1234)
        outs = net.forward(x)
        loss = RL.class_balanced_cross_entropy_loss(outs[-1], gt, size_average=False)
        net.zero_grad()
        loss.backward()
        bwd = {"seed": np.int64(9), "frame_seed": np.int64(1234), "loss_fused_sum": np.float64(loss.item())}
        full, dense, nograd = [], [], []
        for name, p in net.named_parameters():
            if p.grad is None:
                nograd.append(name)
                continue
            m, idx, smp = digest(p.grad, n_samples=4096)
            bwd[f"g_{name}_m"] = m
            if p.grad.numel() <= 65536:
                bwd[f"g_{name}"] = p.grad.numpy().copy()
                full.append(name)
            else:
                bwd[f"g_{name}_i"], bwd[f"g_{name}_s"] = idx, smp
                dense.append(name)
        bwd["full_tensors"], bwd["dense_tensors"], bwd["nograd"] = np.array(full), np.array(dense), np.array(nograd)
        np.savez_compressed(os.path.join(args.out, "bwd_480x854.npz"), **bwd)
    # ------------------------------------------------------------------ 8. fine-tune TRAJECTORY at the BASELINE frame size
    # Section 6's schedule on 1x3x480x854 frames (O.TRAJ_FULL: 30 iterations, step every 5, then the held-out frame): the
    # size north_star's "mask IoU within 1e-3 of the reference" is quoted on.  A few minutes of reference CPU time.
    if not args.skip_e2e and args.only in (None, "trajectory_full"):
        traj = trajectory_fixture(OSVOS_VGG, RL, RNP, O, O.TRAJ_FULL, compact=True)
        np.savez_compressed(os.path.join(args.out, "trajectory_480x854.npz"), **traj)
    print("golden fixtures written to", args.out)


if __name__ == "__main__":
    main()
