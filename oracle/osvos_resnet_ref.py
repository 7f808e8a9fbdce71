"""CPU oracle of the OSVOS_RESNET forward pass - TEST INFRASTRUCTURE, never imported by the product
path (only tests/, __graft_entry__.smoke() and tools/ benchmarks' checker legs use it).

Plain fp32 torch.nn.functional on the CPU, driven by a state_dict with the reference's keys:
  * trunk and head wiring: src/networks/osvos_resnet.py:42-68 (forward), :91-96 (layer_base), :98-121 (stages and
    downsample rule), :124-150 (side_prep / upscale / score_dsn / fuse);
  * the residual blocks are torchvision's (the reference imports them, src/networks/osvos_resnet.py:6-7; its Pipfile
    leaves torchvision unpinned and the package is not in this image), restated from their published definition:
    BasicBlock  = conv3x3(stride) - bn - relu - conv3x3 - bn, + identity or downsample(x), relu;
    Bottleneck  = conv1x1 - bn - relu - conv3x3(stride) - bn - relu - conv1x1(x4) - bn, + residual, relu;
  * centre crop: src/layers/osvos_layers.py:47-54 (floor(d/2) leading pixels dropped).

Pinning: the reference class cannot be instantiated here (torchvision missing - an ordinary ImportError), so this
restatement is pinned by construction only - every op is the same torch CPU operator the reference's modules
dispatch to (F.conv2d, F.batch_norm in eval mode, F.max_pool2d, F.conv_transpose2d), and the state_dict key layout is
checked against the reference's module tree in tests/test_resnet_cpu.py.  The block definitions are "parity unpinned".
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

LAYERS = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottleneck", [3, 4, 6, 3]),
          101: ("bottleneck", [3, 4, 23, 3]), 152: ("bottleneck", [3, 8, 36, 3])}
BN_EPS = 1e-5


def _bn_keys(prefix: str, c: int) -> List[Tuple[str, Tuple[int, ...]]]:
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)), (prefix + ".running_mean", (c,)),
            (prefix + ".running_var", (c,)), (prefix + ".num_batches_tracked", ())]


def state_dict_spec(version: int = 18, scale_down_exponent: int = 0, side_channels: Sequence[int] = None
                    ) -> "OrderedDict[str, Tuple[int, ...]]":
    """Keys and shapes in the registration order of the reference module tree."""
    kind, layers = LAYERS[version]
    exp = 1 if kind == "basic" else 4
    planes = [c // (2 ** scale_down_exponent) for c in (64, 128, 256, 512)]
    spec: List[Tuple[str, Tuple[int, ...]]] = [("layer_base.0.weight", (planes[0], 3, 7, 7))]
    spec += _bn_keys("layer_base.1", planes[0])
    inpl = planes[0]
    for i, (p, nb) in enumerate(zip(planes, layers)):
        for j in range(nb):
            pre = "layer_stages.%d.%d" % (i, j)
            stride = 2 if (i > 0 and j == 0) else 1
            if kind == "basic":
                spec += [(pre + ".conv1.weight", (p, inpl, 3, 3))] + _bn_keys(pre + ".bn1", p)
                spec += [(pre + ".conv2.weight", (p, p, 3, 3))] + _bn_keys(pre + ".bn2", p)
            else:
                spec += [(pre + ".conv1.weight", (p, inpl, 1, 1))] + _bn_keys(pre + ".bn1", p)
                spec += [(pre + ".conv2.weight", (p, p, 3, 3))] + _bn_keys(pre + ".bn2", p)
                spec += [(pre + ".conv3.weight", (4 * p, p, 1, 1))] + _bn_keys(pre + ".bn3", 4 * p)
            if j == 0 and (stride != 1 or inpl != p * exp):
                spec += [(pre + ".downsample.0.weight", (p * exp, inpl, 1, 1))] + _bn_keys(pre + ".downsample.1", p * exp)
            inpl = p * exp
    side_in = list(side_channels) if side_channels is not None else planes
    for i, c in enumerate(side_in):
        spec += [("side_prep.%d.weight" % i, (16, c, 3, 3)), ("side_prep.%d.bias" % i, (16,))]
    for i in range(4):
        spec += [("upscale_side_prep.%d.weight" % i, (16, 16, 2 ** (3 + i), 2 ** (3 + i)))]
    for i in range(4):
        spec += [("score_dsn.%d.weight" % i, (1, 16, 1, 1)), ("score_dsn.%d.bias" % i, (1,))]
    for i in range(4):
        spec += [("upscale_score_dsn.%d.weight" % i, (1, 1, 2 ** (3 + i), 2 ** (3 + i)))]
    spec += [("layer_fuse.weight", (1, 64, 1, 1)), ("layer_fuse.bias", (1,))]
    return OrderedDict(spec)


def bilinear_kernel(size: int) -> np.ndarray:
    """upsample_filt of src/layers/osvos_layers.py:57-66."""
    factor = (size + 1) // 2
    center = factor - 1 if size % 2 == 1 else factor - 0.5
    t = 1 - np.abs(np.arange(size) - center) / factor
    return (t[:, None] * t[None, :]).astype(np.float32)


def make_state_dict(version: int = 18, scale_down_exponent: int = 0, seed: int = 0, trained_head: bool = True,
                    side_channels: Sequence[int] = None) -> "OrderedDict[str, torch.Tensor]":
    """Random but well-conditioned weights: He-scaled convs, BatchNorm with non-trivial affine terms and running
    statistics (so that folding is really exercised), bilinear upscale filters - with small off-diagonal and
    per-channel perturbations when ``trained_head`` (the mimic student trains them, src/mimic.py:74)."""
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, shp in state_dict_spec(version, scale_down_exponent, side_channels).items():
        leaf = k.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            sd[k] = torch.tensor(100, dtype=torch.long)
        elif leaf == "running_mean":
            sd[k] = 0.2 * torch.randn(shp, generator=g)
        elif leaf == "running_var":
            sd[k] = 0.5 + torch.rand(shp, generator=g)
        elif k.startswith("upscale"):
            co = shp[1]
            w = torch.zeros(shp)
            idx = torch.arange(min(shp[0], co))
            w[idx, idx] = torch.from_numpy(bilinear_kernel(shp[2]))
            if trained_head:
                w = w * (1.0 + 0.2 * torch.randn((shp[0], co, 1, 1), generator=g)) + 0.02 * torch.randn(shp, generator=g) / shp[2]
            sd[k] = w
        elif len(shp) == 4:
            fan_in = shp[1] * shp[2] * shp[3]
            sd[k] = torch.randn(shp, generator=g) * (2.0 / fan_in) ** 0.5
        elif ".bn" in k or "downsample.1" in k or k.startswith("layer_base.1"):
            sd[k] = (0.7 + 0.6 * torch.rand(shp, generator=g)) if leaf == "weight" else 0.1 * torch.randn(shp, generator=g)
        else:  # conv biases of the head
            sd[k] = 0.1 * torch.randn(shp, generator=g)
    return sd


def crop_offsets(size: int, target: int) -> Tuple[int, int]:
    d = size - target
    return d // 2, d - d // 2


def center_crop(x: torch.Tensor, height: int, width: int) -> torch.Tensor:
    top, bottom = crop_offsets(x.shape[2], height)
    left, right = crop_offsets(x.shape[3], width)
    return F.pad(x, [-left, -right, -top, -bottom])


def _bn(sd: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor) -> torch.Tensor:
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                        sd[prefix + ".bias"], training=False, eps=BN_EPS)


def _block(sd: Dict[str, torch.Tensor], pre: str, x: torch.Tensor, stride: int) -> torch.Tensor:
    residual = x
    if pre + ".conv3.weight" in sd:
        out = F.relu(_bn(sd, pre + ".bn1", F.conv2d(x, sd[pre + ".conv1.weight"])))
        out = F.relu(_bn(sd, pre + ".bn2", F.conv2d(out, sd[pre + ".conv2.weight"], stride=stride, padding=1)))
        out = _bn(sd, pre + ".bn3", F.conv2d(out, sd[pre + ".conv3.weight"]))
    else:
        out = F.relu(_bn(sd, pre + ".bn1", F.conv2d(x, sd[pre + ".conv1.weight"], stride=stride, padding=1)))
        out = _bn(sd, pre + ".bn2", F.conv2d(out, sd[pre + ".conv2.weight"], padding=1))
    if pre + ".downsample.0.weight" in sd:
        residual = _bn(sd, pre + ".downsample.1", F.conv2d(x, sd[pre + ".downsample.0.weight"], stride=stride))
    return F.relu(out + residual)


def trunk(sd: Dict[str, torch.Tensor], x: torch.Tensor) -> List[torch.Tensor]:
    """The four stage outputs."""
    x = F.relu(_bn(sd, "layer_base.1", F.conv2d(x, sd["layer_base.0.weight"], stride=2, padding=3)))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = []
    for i in range(4):
        j = 0
        while "layer_stages.%d.%d.conv1.weight" % (i, j) in sd:
            x = _block(sd, "layer_stages.%d.%d" % (i, j), x, 2 if (i > 0 and j == 0) else 1)
            j += 1
        outs.append(x)
    return outs


def forward(sd: Dict[str, torch.Tensor], x: torch.Tensor) -> List[torch.Tensor]:
    """[4 side outputs, fused] as OSVOS_RESNET.forward returns them."""
    h, w = int(x.shape[-2]), int(x.shape[-1])
    side, side_out = [], []
    for i, feat in enumerate(trunk(sd, x)):
        prep = F.conv2d(feat, sd["side_prep.%d.weight" % i], sd["side_prep.%d.bias" % i], padding=1)
        f = 2 ** (2 + i)
        side.append(center_crop(F.conv_transpose2d(prep, sd["upscale_side_prep.%d.weight" % i], stride=f), h, w))
        dsn = F.conv2d(prep, sd["score_dsn.%d.weight" % i], sd["score_dsn.%d.bias" % i])
        side_out.append(center_crop(F.conv_transpose2d(dsn, sd["upscale_score_dsn.%d.weight" % i], stride=f), h, w))
    side_out.append(F.conv2d(torch.cat(side, dim=1), sd["layer_fuse.weight"], sd["layer_fuse.bias"]))
    return side_out
