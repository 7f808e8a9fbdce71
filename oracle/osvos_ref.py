"""CPU oracle for the OSVOS-VGG fine-tune hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional (state_dict -> tensors) restatement in plain
PyTorch fp32 of the arithmetic the reference delegates to torch.nn, written to *check*
the HIP path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` may import it; the product path under ``fosvos_amd/`` never does and fails
loudly when the HIP extension is missing.

Parity is PINNED: ``oracle/make_golden.py`` imports the reference itself (in the build
container only) and writes small fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks every function here against them.  The reference has no tests or golden vectors of
its own, so those generated fixtures are the pin.

Reference citations (relative to /root/reference):
  forward            src/networks/osvos_vgg.py:61-83
  stage layout       src/networks/osvos_vgg.py:20-25,85-95
  weight init        src/networks/osvos_vgg.py:97-116
  loss               src/layers/osvos_layers.py:17-44
  centre crop        src/layers/osvos_layers.py:47-54
  bilinear filter    src/layers/osvos_layers.py:57-81
  online SGD groups  src/util/network_provider.py:144-159
  offline SGD groups src/util/network_provider.py:98-125
  online loop body   src/train_online.py:70-101
  offline loop body  src/train_offline.py:77-110
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# Output channels of the 3x3 convs of each stage; stages 1..4 start with a 2x2 ceil-mode
# max-pool (src/networks/osvos_vgg.py:20-25).
STAGE_CHANNELS: Tuple[Tuple[int, ...], ...] = (
    (64, 64), (128, 128), (256, 256, 256), (512, 512, 512), (512, 512, 512))
STAGE_IN: Tuple[int, ...] = (3, 64, 128, 256, 512)
SIDE_CH = 16
BGR_MEAN = (104.00699, 116.66877, 122.67892)  # src/dataloaders/davis_2016.py:28


def conv_module_index(stage: int, k: int) -> int:
    """Index of the k-th conv inside ``stages[stage]`` (an nn.Sequential in the reference).

    Stage 0 is [conv, relu, conv, relu]; stages 1..4 are [pool, conv, relu, ...]
    (src/networks/osvos_vgg.py:85-95)."""
    return 2 * k + (0 if stage == 0 else 1)


def state_dict_spec() -> "OrderedDict[str, Tuple[int, ...]]":
    """The 52 tensors of the reference's state_dict, in registration order
    (src/networks/osvos_vgg.py:50-56): upscale, upscale_, stages, side_prep, score_dsn, fuse."""
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for i in range(4):
        k = 2 ** (i + 2)
        spec[f"upscale.{i}.weight"] = (SIDE_CH, SIDE_CH, k, k)
    for i in range(4):
        k = 2 ** (i + 2)
        spec[f"upscale_.{i}.weight"] = (1, 1, k, k)
    for s, chans in enumerate(STAGE_CHANNELS):
        cin = STAGE_IN[s]
        for j, cout in enumerate(chans):
            m = conv_module_index(s, j)
            spec[f"stages.{s}.{m}.weight"] = (cout, cin, 3, 3)
            spec[f"stages.{s}.{m}.bias"] = (cout,)
            cin = cout
    for i in range(4):
        spec[f"side_prep.{i}.weight"] = (SIDE_CH, STAGE_CHANNELS[i + 1][-1], 3, 3)
        spec[f"side_prep.{i}.bias"] = (SIDE_CH,)
    for i in range(4):
        spec[f"score_dsn.{i}.weight"] = (1, SIDE_CH, 1, 1)
        spec[f"score_dsn.{i}.bias"] = (1,)
    spec["fuse.weight"] = (1, 4 * SIDE_CH, 1, 1)
    spec["fuse.bias"] = (1,)
    return spec


def bilinear_kernel(size: int) -> np.ndarray:
    """Separable bilinear interpolation kernel, float64 (src/layers/osvos_layers.py:57-65)."""
    f = (size + 1) // 2
    c = f - 1.0 if size % 2 == 1 else f - 0.5
    t = 1.0 - np.abs(np.arange(size, dtype=np.float64) - c) / f
    return np.outer(t, t)


def bilinear_deconv_weight(channels: int, size: int) -> torch.Tensor:
    """Diagonal bilinear ConvTranspose2d weight [C, C, k, k] (src/layers/osvos_layers.py:70-81)."""
    w = torch.zeros(channels, channels, size, size, dtype=torch.float32)
    filt = torch.from_numpy(bilinear_kernel(size)).to(torch.float32)
    for c in range(channels):
        w[c, c] = filt
    return w


def make_state_dict(seed: int = 0, scheme: str = "kaiming") -> "OrderedDict[str, torch.Tensor]":
    """Seeded synthetic weights.

    scheme="reference": conv w ~ N(0, 1e-3), b = 0 (src/networks/osvos_vgg.py:99-102).
    scheme="kaiming":   variance-preserving conv weights and small non-zero biases so that
                        activations stay O(input) through 13 layers and parity tests see signal
                        (SURVEY.md §7 step 1).  Deconvs are bilinear in both schemes.
    Tensors are drawn one by one from a CPU torch.Generator, in state_dict order, so the same
    seed gives the same weights on any box with this torch build."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in state_dict_spec().items():
        if name.startswith("upscale"):
            sd[name] = bilinear_deconv_weight(shape[0], shape[2])
            continue
        if scheme == "reference":
            if name.endswith("weight"):
                sd[name] = torch.randn(shape, generator=g) * 1e-3
            else:
                sd[name] = torch.zeros(shape)
            continue
        if scheme != "kaiming":
            raise ValueError(scheme)
        if name.endswith("weight"):
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 2.0 if name.startswith("stages") else 1.0
            sd[name] = torch.randn(shape, generator=g) * math.sqrt(gain / fan_in)
        else:
            sd[name] = torch.randn(shape, generator=g) * 0.1
    return sd


def crop_offsets(size: int, target: int) -> Tuple[int, int]:
    """(leading, trailing) pixels removed by the reference's centre crop for one axis.

    The reference pads by ceil(-d/2) in front and floor(-d/2) behind with d = size - target
    (src/layers/osvos_layers.py:47-54); a negative pad removes pixels, so floor(d/2) go in
    front and ceil(d/2) behind - the odd pixel comes off the bottom/right."""
    d = size - target
    lead = -math.ceil(-d / 2.0)
    trail = -math.floor(-d / 2.0)
    return lead, trail


def center_crop(x: torch.Tensor, height: int, width: int) -> torch.Tensor:
    t, b = crop_offsets(x.shape[2], height)
    l, r = crop_offsets(x.shape[3], width)
    return F.pad(x, [-l, -r, -t, -b])


class _RoundBoth(torch.autograd.Function):
    """bf16 rounding of the value in forward and of the gradient in backward (an activation the HIP
    path stores in bf16 together with its gradient)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


class _RoundValue(torch.autograd.Function):
    """bf16 rounding of the value only (the packed MFMA weight image; its gradient stays fp32)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGrad(torch.autograd.Function):
    """identity in forward, bf16 rounding of the gradient in backward."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float32)


def forward(sd: Dict[str, torch.Tensor], x: torch.Tensor,
            return_intermediates: bool = False, emulate_bf16: bool = False):
    """OSVOS_VGG.forward (src/networks/osvos_vgg.py:61-83): list of 5 logit maps [N,1,H,W].

    emulate_bf16=True re-states the SAME graph with the HIP path's storage precision: conv outputs and
    the gradients flowing through them rounded to bf16, MFMA weight images rounded to bf16, everything
    accumulated in fp32; conv1_1 (fp32 VALU), side_prep outputs and the head stay fp32.  It is the fp32
    reference plus rounding at exactly the points where the kernels round, which separates "precision
    scheme" from "kernel correctness" in the parity tests."""
    H, W = int(x.shape[-2]), int(x.shape[-1])
    rb = _RoundBoth.apply if emulate_bf16 else (lambda t: t)
    rv = _RoundValue.apply if emulate_bf16 else (lambda t: t)
    rg = _RoundGrad.apply if emulate_bf16 else (lambda t: t)
    feats: List[torch.Tensor] = []
    sides: List[torch.Tensor] = []
    side_out: List[torch.Tensor] = []
    side_prep_out: List[torch.Tensor] = []
    h = x
    for s, chans in enumerate(STAGE_CHANNELS):
        if s > 0:
            h = rg(F.max_pool2d(h, kernel_size=2, stride=2, ceil_mode=True))
        for j in range(len(chans)):
            m = conv_module_index(s, j)
            w = sd[f"stages.{s}.{m}.weight"]
            if not (s == 0 and j == 0):
                w = rv(w)  # conv1_1 runs on fp32 weights and the fp32 frame
            h = rb(F.relu(F.conv2d(h, w, sd[f"stages.{s}.{m}.bias"], padding=1)))
        feats.append(h)
        if s == 0:
            continue
        i = s - 1
        stride = 2 ** s
        prep = rg(F.conv2d(rg(h), rv(sd[f"side_prep.{i}.weight"]), sd[f"side_prep.{i}.bias"], padding=1))
        side_prep_out.append(prep)
        up = F.conv_transpose2d(prep, sd[f"upscale.{i}.weight"], stride=stride)
        sides.append(center_crop(up, H, W))
        score = F.conv2d(prep, sd[f"score_dsn.{i}.weight"], sd[f"score_dsn.{i}.bias"])
        up1 = F.conv_transpose2d(score, sd[f"upscale_.{i}.weight"], stride=stride)
        side_out.append(center_crop(up1, H, W))
    fused = F.conv2d(torch.cat(sides, dim=1), sd["fuse.weight"], sd["fuse.bias"])
    outs = side_out + [fused]
    if return_intermediates:
        return outs, {"feats": feats, "side_prep": side_prep_out}
    return outs


def cbce_loss(output: torch.Tensor, label: torch.Tensor, size_average: bool = True) -> torch.Tensor:
    """Class-balanced BCE with logits (src/layers/osvos_layers.py:17-44).

    y = label >= 0.5;  per-pixel l = softplus-stable BCE(x, y);
    L = (Nn/N) * sum_{y=1} l + (Np/N) * sum_{y=0} l;  divided by numel if size_average."""
    y = (label >= 0.5).to(output.dtype)
    n_pos = y.sum()
    n_neg = (1.0 - y).sum()
    n_tot = n_pos + n_neg
    # l = max(x,0) - x*y + log(1 + exp(-|x|)), with the reference's ">= 0" indicator choosing
    # the branch, so that autograd at x == 0 gives sigmoid(0) - y exactly as the reference does.
    nonneg = output >= 0
    per_px = (torch.where(nonneg, output, torch.zeros_like(output)) - output * y
              + torch.log(1 + torch.exp(torch.where(nonneg, -output, output))))
    loss_pos = (y * per_px).sum()
    loss_neg = ((1.0 - y) * per_px).sum()
    loss = n_neg / n_tot * loss_pos + n_pos / n_tot * loss_neg
    if size_average:
        loss = loss / float(label.numel())
    return loss


def cbce_loss_grad(output: torch.Tensor, label: torch.Tensor, size_average: bool = True) -> torch.Tensor:
    """Closed-form dL/dx = w_i (sigmoid(x_i) - y_i), w_i = Nn/N if y_i = 1 else Np/N."""
    y = (label >= 0.5).to(output.dtype)
    n_pos = y.sum()
    n_tot = float(y.numel())
    n_neg = n_tot - n_pos
    w = torch.where(y > 0, n_neg / n_tot, n_pos / n_tot)
    g = w * (torch.sigmoid(output) - y)
    if size_average:
        g = g / n_tot
    return g


def mask_iou(a: torch.Tensor, b: torch.Tensor) -> float:
    """IoU of two boolean masks, |A and B| / |A or B| (1.0 when both empty).  The reference
    delegates evaluation to the external DAVIS toolkit (src/eval/README.md:1-3)."""
    a = a.bool()
    b = b.bool()
    union = (a | b).sum().item()
    if union == 0:
        return 1.0
    return (a & b).sum().item() / union


def logits_to_mask(x: torch.Tensor) -> torch.Tensor:
    """sigmoid(x) >= 0.5  <=>  x >= 0 (src/run_webcam.py:91-93)."""
    return x >= 0


# ----------------------------------------------------------------------------------------------
# Optimizer recipe
# ----------------------------------------------------------------------------------------------
def _split(params: Dict[str, torch.Tensor], prefix: str, kind: str) -> List[torch.Tensor]:
    return [p for n, p in params.items() if n.startswith(prefix + ".") and kind in n[len(prefix):]]


def sgd_param_groups(params: Dict[str, torch.Tensor], mode: str = "online", lr: float = 1e-8,
                     weight_decay: float = 0.0002) -> List[dict]:
    """Param groups of VGGOnlineProvider.get_optimizer (src/util/network_provider.py:144-159)
    or VGGOfflineProvider.get_optimizer (:98-125).  ``params`` maps state_dict names to leaf tensors."""
    groups: List[dict] = [
        {"params": _split(params, "stages", "weight"), "weight_decay": weight_decay},
        {"params": _split(params, "stages", "bias"), "lr": 2 * lr},
        {"params": _split(params, "side_prep", "weight"), "weight_decay": weight_decay},
        {"params": _split(params, "side_prep", "bias"), "lr": 2 * lr},
    ]
    if mode == "offline":
        groups += [
            {"params": _split(params, "score_dsn", "weight"), "lr": lr / 10, "weight_decay": weight_decay},
            {"params": _split(params, "score_dsn", "bias"), "lr": 2 * lr / 10},
        ]
    elif mode != "online":
        raise ValueError(mode)
    groups += [
        {"params": _split(params, "upscale", "weight"), "lr": 0},
        {"params": _split(params, "upscale_", "weight"), "lr": 0},
        {"params": [params["fuse.weight"]], "lr": lr / 100, "weight_decay": weight_decay},
        {"params": [params["fuse.bias"]], "lr": 2 * lr / 100},
    ]
    return groups


def make_sgd(params: Dict[str, torch.Tensor], mode: str = "online", lr: float = 1e-8,
             weight_decay: float = 0.0002, momentum: float = 0.9) -> torch.optim.SGD:
    return torch.optim.SGD(sgd_param_groups(params, mode, lr, weight_decay), lr=lr, momentum=momentum)


def leaf_params(sd: Dict[str, torch.Tensor]) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())


def online_loop(sd: Dict[str, torch.Tensor], images: Sequence[torch.Tensor], gts: Sequence[torch.Tensor],
                n_iters: int, avg_grad_every_n: int = 5, lr: float = 1e-8):
    """The body of train_online._train (src/train_online.py:70-101): fwd -> loss on outputs[-1]
    (size_average=False) -> /avg_grad_every_n -> backward -> step + zero_grad every n-th.
    images/gts are cycled.  Returns (per-iteration unscaled losses, final params)."""
    params = leaf_params(sd)
    opt = make_sgd(params, "online", lr=lr)
    losses: List[float] = []
    counter = 0
    for it in range(n_iters):
        x = images[it % len(images)]
        y = gts[it % len(gts)]
        outs = forward(params, x)
        loss = cbce_loss(outs[-1], y, size_average=False)
        losses.append(float(loss.item()))
        (loss / avg_grad_every_n).backward()
        counter += 1
        if counter % avg_grad_every_n == 0:
            opt.step()
            opt.zero_grad()
            counter = 0
    return losses, OrderedDict((k, v.detach()) for k, v in params.items())


def offline_loop(sd: Dict[str, torch.Tensor], images: Sequence[torch.Tensor], gts: Sequence[torch.Tensor],
                 n_iters: int, epoch: int = 0, n_epochs: int = 240, avg_grad_every_n: int = 10,
                 lr: float = 1e-8):
    """The body of train_offline._train (src/train_offline.py:77-110): 5 deeply-supervised losses,
    loss = (1 - epoch/n_epochs) * sum(side losses) + fused loss, /avg_grad_every_n, step every n-th."""
    params = leaf_params(sd)
    opt = make_sgd(params, "offline", lr=lr)
    trace: List[List[float]] = []
    counter = 0
    for it in range(n_iters):
        x = images[it % len(images)]
        y = gts[it % len(gts)]
        outs = forward(params, x)
        ls = [cbce_loss(o, y, size_average=False) for o in outs]
        trace.append([float(l.item()) for l in ls])
        loss = (1 - epoch / n_epochs) * sum(ls[:-1]) + ls[-1]
        (loss / avg_grad_every_n).backward()
        counter += 1
        if counter % avg_grad_every_n == 0:
            opt.step()
            opt.zero_grad()
            counter = 0
    return trace, OrderedDict((k, v.detach()) for k, v in params.items())


# ----------------------------------------------------------------------------------------------
# Synthetic inputs: uniform BGR frame minus the dataset mean, elliptical mask
# ----------------------------------------------------------------------------------------------
def synthetic_frame(n: int, h: int, w: int, seed: int = 1234) -> Tuple[torch.Tensor, torch.Tensor]:
    """Uniform-noise BGR frame with a brighter, lower-contrast elliptical object (~10 % of the pixels),
    minus the dataset mean; gt = the ellipse.  The object is photometrically distinct so that a network
    can actually learn it (needed for the confident-logit IoU tests)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    noise = torch.rand((n, 3, h, w), generator=g)
    yy = torch.arange(h, dtype=torch.float32).view(h, 1)
    xx = torch.arange(w, dtype=torch.float32).view(1, w)
    gt = torch.zeros((n, 1, h, w), dtype=torch.float32)
    for i in range(n):
        cy, cx = h * (0.45 + 0.05 * i), w * (0.5 - 0.03 * i)
        ry, rx = h * 0.2, w * 0.16  # pi*0.2*0.16 ~= 10 % foreground
        gt[i, 0] = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0).float()
    img = gt * (150.0 + 100.0 * noise) + (1.0 - gt) * (150.0 * noise)
    img = img - torch.tensor(BGR_MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    return img, gt


# ----------------------------------------------------------------------------------------------
# Fine-tune trajectory fixture (tests/golden/trajectory.npz, written by oracle/make_golden.py section 6 with the
# reference's own modules): the schedule and its inputs, seeds only, so every box rebuilds the same tensors
# ----------------------------------------------------------------------------------------------
TRAJ = {"seed": 41, "head_scale": 0.02, "lr": 1e-7, "iters": 60, "avg": 5, "h": 96, "w": 160, "frame_seed": 301,
        "heldout_seed": 302}


# The same schedule at the frame size BASELINE.json quotes the metric on (tests/golden/trajectory_480x854.npz, section 7 of
# oracle/make_golden.py).  The loss is a SUM over pixels (size_average=False, src/train_online.py:81), so the learning rate
# that keeps the run smooth shrinks with the pixel count (409,920 against 15,360 pixels).
TRAJ_FULL = {"seed": 41, "head_scale": 0.02, "lr": 4e-9, "iters": 30, "avg": 5, "h": 480, "w": 854, "frame_seed": 301,
             "heldout_seed": 302}


def trajectory_inputs(T=None):
    """(parent state_dict, training frames [(x, gt)], held-out (x, gt)).  The parent is the seeded Kaiming net with its
    fuse weights scaled down (logits start O(1), so the run is smooth); the training frames are the annotated frame and
    its horizontal flip (src/dataloaders/custom_transforms.py:95-106 with the draw fixed); the held-out frame shows the
    object displaced.  T: TRAJ (default) or TRAJ_FULL."""
    T = TRAJ if T is None else T
    sd = make_state_dict(T["seed"])
    sd["fuse.weight"] = sd["fuse.weight"] * T["head_scale"]
    x, gt = synthetic_frame(1, T["h"], T["w"], seed=T["frame_seed"])
    frames = [(x, gt), (x.flip(3).contiguous(), gt.flip(3).contiguous())]
    xh, gh = synthetic_frame(2, T["h"], T["w"], seed=T["heldout_seed"])
    return sd, frames, (xh[1:].contiguous(), gh[1:].contiguous())
