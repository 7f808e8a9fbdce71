"""Layer loop of the OSVOS_RESNET inference path over the C ABI: pack (BatchNorm folded) once per
weight version, then one kernel per conv - residual add and ReLU included - plus the pool and the head.

The walk is driven by the module tree, not by a version table: any trunk made of blocks with conv1/bn1/conv2/bn2
[/conv3/bn3] and an optional ``downsample`` Sequential(conv, bn) runs, which covers BasicBlock, Bottleneck and the
BasicBlockDummy nets of src/prune.py whatever their (pruned) channel counts.  Nothing here computes on the CPU and
nothing imports the oracle; a module the kernels cannot express raises.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

import ctypes
import os

from . import Context, Conv2dDesc, ResnetBlock, ResnetNet, check, lib, ops, ptr_array4


def use_mfma() -> bool:
    """FOSVOS_RESNET_MFMA=0 keeps every layer on the vector-ALU kernel (A/B runs)."""
    return os.environ.get("FOSVOS_RESNET_MFMA", "1") != "0"


def width(c: int) -> int:
    """Channels a feature map is STORED with.  On the MFMA path every map is widened with zero channels to a count the
    implicit GEMM takes on both sides (32, or a multiple of 64): a 16-, 8- or 59-channel layer then runs on the matrix
    cores - padded weights, BatchNorm terms and biases are zero, so the extra channels hold exact zeros end to end and
    the arithmetic on the real ones is unchanged.  Without MFMA: the next multiple of 8 (16-byte vectors)."""
    if not use_mfma():
        return (c + 7) // 8 * 8
    return 32 if c <= 32 else (c + 63) // 64 * 64


def _pad_to(t: Optional[torch.Tensor], n: int, value: float = 0.0) -> Optional[torch.Tensor]:
    if t is None or t.shape[0] == n:
        return t
    out = torch.full((n,) + tuple(t.shape[1:]), value, dtype=t.dtype, device=t.device)
    out[:t.shape[0]] = t
    return out


class _Conv:
    """One packed conv (+ folded BatchNorm) between maps of physical widths ci_p -> co_p (see width()).  3x3 layers whose
    physical channel counts fit the MFMA implicit GEMM (Ci % 32 == 0; Co % 64 == 0, Co = 32, or the 16-channel side_prep
    form) take that path (kind 1), everything else the vector-ALU direct conv (kind 0)."""
    __slots__ = ("packed", "bias", "ci", "co", "k", "stride", "kind", "ci_real", "co_real")

    def __init__(self, conv: nn.Conv2d, bn: Optional[nn.BatchNorm2d], what: str, ci_p: int, co_p: int,
                 allow_mfma: bool = True) -> None:
        k = conv.kernel_size[0]
        if (conv.kernel_size not in ((1, 1), (3, 3)) or conv.stride not in ((1, 1), (2, 2)) or conv.dilation != (1, 1)
                or conv.groups != 1 or conv.padding != (k // 2, k // 2)):
            raise NotImplementedError(f"{what}: {conv} - the HIP path has k in (1, 3), stride in (1, 2), padding k // 2")
        if bn is not None and (not bn.track_running_stats or bn.running_mean is None or not bn.affine):
            raise NotImplementedError(f"{what}: BatchNorm without affine running statistics cannot be folded")
        self.ci_real, self.co_real = conv.in_channels, conv.out_channels
        self.ci, self.co, self.k, self.stride = ci_p, co_p, k, conv.stride[0]
        w = conv.weight.detach()
        if (co_p, ci_p) != tuple(w.shape[:2]):
            wp = torch.zeros((co_p, ci_p, k, k), dtype=w.dtype, device=w.device)
            wp[:w.shape[0], :w.shape[1]] = w
            w = wp
        bnp = None
        if bn is not None:  # padded channels: scale 0 (weight 0 over variance 1), shift 0
            bnp = (_pad_to(bn.weight.detach(), co_p), _pad_to(bn.bias.detach(), co_p), _pad_to(bn.running_mean, co_p),
                   _pad_to(bn.running_var, co_p, 1.0), bn.eps)
        cb = None if conv.bias is None else _pad_to(conv.bias.detach(), co_p)
        # (16 output channels is the side_prep shape: the MFMA path has a 16-wide tile with an fp32 store for it;
        # stride 2 runs there too, with a subsampling store: fosvos_conv3x3_s2_fwd)
        self.kind = int(allow_mfma and use_mfma() and k == 3 and ci_p % 32 == 0
                        and (co_p % 64 == 0 or co_p == 32 or (co_p == 16 and self.stride == 1)))
        if self.kind:
            folded, self.bias = ops.fold_conv_bn(w.contiguous(), cb, bnp)
            self.packed, _ = ops.pack_conv3x3_weights(folded, want_fwd=True, want_dgrad=False)
        else:
            self.packed, self.bias = ops.pack_conv2d_bn(w.contiguous(), cb, bnp)

    def __call__(self, x: torch.Tensor, relu: bool, addend: Optional[torch.Tensor] = None,
                 out_f32: bool = False) -> torch.Tensor:
        if self.kind and self.co == 16:
            if addend is not None:
                raise NotImplementedError("16-channel MFMA conv has no residual epilogue")
            return ops.conv3x3_fwd_add(x, self.packed, self.bias, self.ci, self.co, relu, None, out_f32=out_f32)
        if self.kind and self.stride == 2:
            if addend is not None:
                raise NotImplementedError("stride-2 MFMA conv has no residual epilogue")
            return ops.conv3x3_s2_fwd(x, self.packed, self.bias, self.ci, self.co, relu)
        if self.kind:
            return ops.conv3x3_fwd_add(x, self.packed, self.bias, self.ci, self.co, relu, addend)
        return ops.conv2d_fwd(x, self.packed, self.bias, self.ci, self.co, self.k, self.stride, relu, addend, out_f32)

    def desc(self) -> Conv2dDesc:
        return Conv2dDesc(self.packed.data_ptr(), self.bias.data_ptr(), self.ci, self.co, self.k, self.stride, self.kind)


class _Block:
    __slots__ = ("convs", "down", "c_out")

    def __init__(self, blk: nn.Module, what: str, c_in: int) -> None:
        pairs = [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2)]
        if hasattr(blk, "conv3"):
            pairs.append((blk.conv3, blk.bn3))
        self.convs = []
        c = c_in
        for i, (conv, bn) in enumerate(pairs):
            if conv.in_channels != c:
                raise RuntimeError(f"{what}.conv{i + 1} expects {conv.in_channels} input channels, its input has {c}")
            if bn.num_features != conv.out_channels:
                raise RuntimeError(f"{what}.bn{i + 1} has {bn.num_features} features for {conv.out_channels} channels")
            self.convs.append(_Conv(conv, bn, f"{what}.conv{i + 1}", width(c), width(conv.out_channels)))
            c = conv.out_channels
        self.c_out = c
        self.down = None
        if blk.downsample is not None:
            dconv, dbn = blk.downsample[0], blk.downsample[1]
            if dconv.in_channels != c_in or dconv.out_channels != c:
                raise RuntimeError(f"{what}.downsample maps {dconv.in_channels} -> {dconv.out_channels} channels, the block "
                                   f"{c_in} -> {c}")
            self.down = _Conv(dconv, dbn, f"{what}.downsample", width(c_in), width(c), allow_mfma=False)
        elif c != c_in:
            raise RuntimeError(f"{what}: identity residual with {c_in} channels in and {c} out")

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        res = x if self.down is None else self.down(x, relu=False)
        y = x
        for c in self.convs[:-1]:
            y = c(y, relu=True)
        return self.convs[-1](y, relu=True, addend=res)  # relu(bn(conv(y)) + residual)


class ResnetPlan:
    """Device-side images of one net's weights; rebuilt when any parameter or buffer was written or moved."""

    def __init__(self) -> None:
        self.signature = None
        self.aux = None  # the auxiliary HIP stream of the native loop (created on first use)

    @staticmethod
    def _signature(net: nn.Module):
        return (use_mfma(), os.environ.get("FOSVOS_RESNET_FUSE_FIRST", "1")) + tuple(
            (t.data_ptr(), t._version) for t in list(net.parameters()) + list(net.buffers()))

    # Walking the module tree for that signature costs 0.11-0.19 ms per call on the GPU box's host - as much as the device
    # needs for a whole 1080p frame of a thinned net.  So the walk is done once per packing; every later call re-checks the
    # recorded path instead: each module on it is still the child of its parent (a pruned block swapped in is seen), each
    # tensor is still the attribute of its module, sits at the same address and has the same version counter.
    def _record(self, net: nn.Module) -> None:
        links, params, bufs = [], [], []

        def walk(m):
            for name, t in m._parameters.items():
                if t is not None:
                    params.append((m._parameters, name, t, t.data_ptr(), t._version))
            for name, t in m._buffers.items():
                if t is not None:
                    bufs.append((m._buffers, name, t, t.data_ptr(), t._version))
            for name, child in m._modules.items():
                if child is not None:
                    links.append((m._modules, name, child))
                    walk(child)
            links.append((m._modules, None, len(m._modules)))

        walk(net)
        self._links, self._watched = links, params + bufs

    def _unchanged(self) -> bool:
        if self.signature is None or self.signature[:2] != (use_mfma(), os.environ.get("FOSVOS_RESNET_FUSE_FIRST", "1")):
            return False
        for d, name, child in self._links:
            if (len(d) != child) if name is None else (d.get(name) is not child):
                return False
        for d, name, t, ptr, ver in self._watched:
            if d.get(name) is not t or t._version != ver or t.data_ptr() != ptr:
                return False
        return True

    def refresh(self, net: nn.Module) -> None:
        if self._unchanged():
            return
        sig = self._signature(net)
        conv1, bn1, _relu, pool = net.layer_base[0], net.layer_base[1], net.layer_base[2], net.layer_base[3]
        if ((conv1.kernel_size, conv1.stride, conv1.padding, conv1.in_channels) != ((7, 7), (2, 2), (3, 3), 3)
                or conv1.bias is not None):
            raise NotImplementedError(f"layer_base conv {conv1}: the HIP path has the 7x7 stride-2 conv on 3-channel frames")
        if (pool.kernel_size, pool.stride, pool.padding, pool.ceil_mode) != (3, 2, 1, False):
            raise NotImplementedError(f"layer_base pool {pool}: the HIP path has MaxPool2d(3, 2, 1)")
        c = conv1.out_channels
        self.c0 = width(c)  # (the first layer writes the padded width too: zero filters, zero shift)
        w1 = conv1.weight.detach()
        if self.c0 != c:
            w1p = torch.zeros((self.c0, 3, 7, 7), dtype=w1.dtype, device=w1.device)
            w1p[:c] = w1
            w1 = w1p
        self.first = ops.pack_conv7x7_bn(w1.contiguous(),
                                         (_pad_to(bn1.weight.detach(), self.c0), _pad_to(bn1.bias.detach(), self.c0),
                                          _pad_to(bn1.running_mean, self.c0), _pad_to(bn1.running_var, self.c0, 1.0), bn1.eps))
        if len(net.layer_stages) != 4 or len(net.side_prep) != 4:
            raise NotImplementedError("the side-output head takes exactly four stages")
        self.stages: List[List[_Block]] = []
        self.side: List[_Conv] = []
        for i, stage in enumerate(net.layer_stages):
            blocks = []
            for j, b in enumerate(stage):
                blocks.append(_Block(b, f"layer_stages.{i}.{j}", c))
                c = blocks[-1].c_out
            self.stages.append(blocks)
            sp = net.side_prep[i]
            if sp.in_channels != c:
                raise RuntimeError(f"side_prep.{i} expects {sp.in_channels} input channels, the stage produces {c}")
            self.side.append(_Conv(sp, None, f"side_prep.{i}", width(c), sp.out_channels))
        fuse_w = net.layer_fuse.weight.detach()
        if tuple(fuse_w.shape) != (1, 64, 1, 1) or any(s.co != 16 for s in self.side):
            raise NotImplementedError("the head kernel is built for 16-channel side maps and one output channel")
        self.strides, self.filt, self.filt1 = [], [], []
        for s in range(4):
            up, up1 = net.upscale_side_prep[s], net.upscale_score_dsn[s]
            f = up.stride[0]
            for m in (up, up1):
                if m.kernel_size != (2 * f, 2 * f) or m.stride != (f, f) or m.padding != (0, 0) or m.bias is not None:
                    raise NotImplementedError(f"{m}: the head kernel has kernel = 2 x stride, no padding, no bias")
            if tuple(up.weight.shape[:2]) != (16, 16) or tuple(up1.weight.shape[:2]) != (1, 1):
                raise NotImplementedError("upscale layers must be 16 -> 16 and 1 -> 1")
            # transposed conv and 1x1 fuse are both linear: contract them into one [k][k][16] filter per scale
            self.filt.append(torch.einsum("o,iokl->kli", fuse_w[0, 16 * s:16 * s + 16, 0, 0], up.weight.detach()).contiguous())
            self.filt1.append(up1.weight.detach()[0, 0].contiguous())
            self.strides.append(f)
        self.dsn_w = torch.cat([m.weight.detach().reshape(1, 16) for m in net.score_dsn]).contiguous()
        self.dsn_b = torch.cat([m.bias.detach().reshape(1) for m in net.score_dsn]).contiguous()
        self.fuse_b = net.layer_fuse.bias.detach().reshape(1).contiguous()
        self._build_native()
        self._record(net)
        self.signature = sig

    def _build_native(self) -> None:
        """The host structs fosvos_resnet_forward walks (they point into the tensors this plan keeps alive)."""
        flat = [b for stage in self.stages for b in stage]
        self.c_blocks = (ResnetBlock * len(flat))()
        for cb, b in zip(self.c_blocks, flat):
            cb.n_convs = len(b.convs)
            for q, c in enumerate(b.convs):
                cb.conv[q] = c.desc()
            cb.has_down = 0 if b.down is None else 1
            if b.down is not None:
                cb.down = b.down.desc()
        net = ResnetNet()
        net.first_w, net.first_b, net.first_co = self.first[0].data_ptr(), self.first[1].data_ptr(), self.c0
        net.first_fp32_math = 0 if use_mfma() else 1
        net.first_unfused = int(os.environ.get("FOSVOS_RESNET_FUSE_FIRST", "1") == "0")
        for s in range(4):
            net.blocks_per_stage[s] = len(self.stages[s])
            net.side[s] = self.side[s].desc()
            net.filt[s], net.filt1[s], net.stride[s] = self.filt[s].data_ptr(), self.filt1[s].data_ptr(), self.strides[s]
        net.blocks = ctypes.cast(self.c_blocks, ctypes.POINTER(ResnetBlock))
        net.dsn_w, net.dsn_b, net.fuse_b = self.dsn_w.data_ptr(), self.dsn_b.data_ptr(), self.fuse_b.data_ptr()
        self.c_net = net
        self.arena = None


def _checked_input(net: nn.Module, x: torch.Tensor) -> torch.Tensor:
    if net.training:
        raise RuntimeError("OSVOS_RESNET runs eval-mode inference on the HIP path (BatchNorm folded from its running "
                           "statistics); training-mode forward/backward is not built - call net.eval()")
    if not x.is_cuda:
        raise RuntimeError("OSVOS_RESNET.forward: the input must live on the GPU (the HIP path has no CPU fallback)")
    if x.dim() != 4 or x.shape[1] != 3:
        raise ValueError(f"OSVOS_RESNET.forward: expected [N,3,H,W], got {tuple(x.shape)}")
    return x.detach().contiguous().float()


def forward_ops(net: nn.Module, plan: ResnetPlan, x: torch.Tensor) -> List[torch.Tensor]:
    """The layer loop in Python, one C-ABI call per kernel (per-kernel timing with ops.OpProfiler, and the reference
    point the native loop is tested against)."""
    x = _checked_input(net, x)
    with torch.no_grad():
        plan.refresh(net)
        n, _c, h, w = x.shape
        if use_mfma() and plan.c0 <= 32 and os.environ.get("FOSVOS_RESNET_FUSE_FIRST", "1") != "0":
            y = ops.conv7x7s2_pool_first_fwd(x, plan.first[0], plan.first[1], plan.c0)   # conv + pool, one launch
        else:
            y = ops.conv7x7s2_first_fwd(x, plan.first[0], plan.first[1], plan.c0, relu=True, fp32_math=not use_mfma())
            y = ops.maxpool3x3s2_fwd(y)
        sides = []
        for blocks, side in zip(plan.stages, plan.side):
            for blk in blocks:
                y = blk(y)
            sides.append(side(y, relu=False, out_f32=True))
        fused, outs = ops.deconv_head_fwd(sides, plan.strides, plan.filt, plan.filt1, plan.dsn_w, plan.dsn_b, plan.fuse_b,
                                          h, w, with_side_out=True)
    return outs + [fused]


def forward(net: nn.Module, plan: ResnetPlan, x: torch.Tensor) -> List[torch.Tensor]:
    """[4 side outputs, fused], each [N,1,H,W] fp32 logits (src/networks/osvos_resnet.py:42-68): one native call
    issues every kernel of the pass."""
    x = _checked_input(net, x)
    plan.refresh(net)
    n, _c, h, w = x.shape
    L = lib()
    need = L.fosvos_resnet_arena_bytes(ctypes.byref(plan.c_net), n, h, w)
    if need == 0:
        msg = L.fosvos_last_error().decode(errors="replace")
        raise RuntimeError(f"OSVOS_RESNET.forward: {msg}")
    if plan.arena is None or plan.arena.numel() < need or plan.arena.device != x.device:
        plan.arena = torch.empty(need, dtype=torch.uint8, device=x.device)
    # net.compute_side_outputs = False (a caller that reads outputs[-1] only, like the reference's test loop,
    # src/util/experiment_helper.py:49): the head skips the four side logit maps, empty placeholders take their place
    with_side = bool(getattr(net, "compute_side_outputs", True))
    outs = [torch.empty((n, 1, h, w) if with_side or i == 4 else (0,), dtype=torch.float32, device=x.device) for i in range(5)]
    dev = x.device.index if x.device.index is not None else torch.cuda.current_device()
    aux = ctx = None
    # FOSVOS_RESNET_AUX=1 issues the side_prep / downsample convs on a second stream beside the trunk.  Off by default:
    # measured at 1080p it LOSES 0.09-0.12 ms per frame on every net (the ~20 cross-stream event waits cost more than the
    # seven small kernels they take off the chain).
    if os.environ.get("FOSVOS_RESNET_AUX", "0") == "1":
        if plan.aux is None or plan.aux.device != x.device:
            plan.aux = torch.cuda.Stream(device=x.device)
        aux = plan.aux.cuda_stream
        if getattr(plan, "ctx", None) is None or plan.ctx.device != dev:
            plan.ctx = Context(dev)  # the events that order the two streams belong to this model's plan
        ctx = plan.ctx.handle
    check(L.fosvos_resnet_forward(ctypes.byref(plan.c_net), x.data_ptr(), n, h, w, plan.arena.data_ptr(),
                                  plan.arena.numel(), outs[4].data_ptr(),
                                  ptr_array4([o.data_ptr() if with_side else None for o in outs[:4]]), dev,
                                  torch.cuda.current_stream(dev).cuda_stream, ctx, aux), "resnet_forward")
    return outs
