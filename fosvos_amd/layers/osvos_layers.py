"""Drop-in for the reference's ``layers.osvos_layers`` (src/layers/osvos_layers.py): same function
names, argument meaning and return values; the loss runs as one fused HIP kernel group.

  class_balanced_cross_entropy_loss  src/layers/osvos_layers.py:17-44  -> fosvos_cbce_loss
  class_balanced_cross_entropy_loss_frames  (extension) the same loss per frame of a batch
  center_crop                        src/layers/osvos_layers.py:47-54  (index arithmetic only)
  upsample_filt / interp_surgery     src/layers/osvos_layers.py:57-81  (one-off host init)
  logit / sigmoid_np                 src/layers/osvos_layers.py:9-14   (numpy helpers)
"""
from __future__ import division

import math

import numpy as np
import torch
from torch.nn import functional as F

from fosvos_hip import ops


def logit(x):
    return np.log(x / (1 - x + 1e-08) + 1e-08)


def sigmoid_np(x):
    return 1 / (1 + np.exp(-x))


def _seed_of(backward_seed):
    """(data_ptr, value) of an announced backward seed, (None, 1.0) without one."""
    if backward_seed is None:
        return None, 1.0
    tensor, value = backward_seed
    return tensor.data_ptr(), float(value)


def _scaled(ctx, g):
    """d(loss)/d(logits) times the incoming gradient.  With an announced seed the kernel already wrote the gradient times
    that value: the very tensor that was announced needs no multiplication, anything else is divided by the value first."""
    global seed_hits
    if ctx.seed_ptr is None:
        return ctx.grad * g.reshape(-1, *([1] * (ctx.grad.dim() - 1))) if g.dim() else ctx.grad * g
    if g.data_ptr() == ctx.seed_ptr:
        seed_hits += 1
        return ctx.grad
    g = g / ctx.seed_value
    return ctx.grad * g.reshape(-1, *([1] * (ctx.grad.dim() - 1))) if g.dim() else ctx.grad * g


seed_hits = 0  # backward passes that found their gradient already scaled (tests read it)


class _CBCELoss(torch.autograd.Function):
    """Loss value and d(loss)/d(logits) come out of the same kernel pass; backward only scales."""

    @staticmethod
    def forward(ctx, output, label, size_average, batch_counts=None, backward_seed=None):
        ctx.seed_ptr, ctx.seed_value = _seed_of(backward_seed)
        loss, grad = ops.cbce_loss(output.contiguous().float(), label.contiguous().float(),
                                   size_average=bool(size_average), grad_scale=ctx.seed_value,
                                   want_grad=output.requires_grad, batch_counts=batch_counts)
        ctx.grad = grad
        ctx.shape = output.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        if ctx.grad is None:
            return None, None, None, None, None
        return _scaled(ctx, g).reshape(ctx.shape), None, None, None, None


def class_balanced_cross_entropy_loss(output, label, size_average=True, batch_counts=None, backward_seed=None):
    """Class-balanced cross entropy loss (same contract as the reference).

    Args:
    output: Output of the network (logits)
    label: Ground truth label
    batch_counts: (extension, data-parallel training only) float64 [2] device tensor {positives, pixels} of the WHOLE
        batch when ``output`` is one rank's shard of it (``parallel.batch_label_counts``): the reference balances the
        classes over the batch tensor (src/layers/osvos_layers.py:28-39), so a shard needs the batch's counts for the
        ranks' losses to add up to the single-process value.  None: count ``label`` itself (the reference's behaviour).
    backward_seed: (extension) ``(tensor, value)`` - the caller announces that it will call ``loss.backward(tensor)`` and
        that every element of ``tensor`` equals ``value`` (the online loop's 1 / nAveGrad): the loss kernel then writes the
        gradient already multiplied, and the backward pass of that very tensor launches nothing.  Any other incoming
        gradient is still handled correctly (divided by ``value`` first).
    Returns:
    0-dim tensor with the loss; reductions run over the whole batch tensor."""
    if not output.is_cuda:
        raise RuntimeError("class_balanced_cross_entropy_loss: the HIP implementation needs GPU tensors "
                           "(no CPU fallback)")
    if label.device != output.device:
        label = label.to(output.device)
    if tuple(label.shape) != tuple(output.shape):
        raise ValueError("class_balanced_cross_entropy_loss: output {} vs label {}".format(
            tuple(output.shape), tuple(label.shape)))
    return _CBCELoss.apply(output, label, size_average, batch_counts, backward_seed)


class _CBCELossFrames(torch.autograd.Function):
    """The loss of every frame of a batch on its own ([N,1,H,W] -> [N]): one set of launches (fosvos_cbce_loss_frames),
    gradients in one buffer so that backward is a single scale."""

    @staticmethod
    def forward(ctx, output, label, size_average, backward_seed=None, staged=None):
        ctx.seed_ptr, ctx.seed_value = _seed_of(backward_seed)
        if staged is not None:  # the class counts are already in the staged loss's workspace; the values come with finish()
            losses, grad = staged.loss(output.contiguous().float(), size_average=bool(size_average),
                                       want_grad=output.requires_grad, grad_scale=ctx.seed_value)
        else:
            losses, grad = ops.cbce_loss_frames(output.contiguous().float(), label.contiguous().float(),
                                                size_average=bool(size_average), want_grad=output.requires_grad,
                                                grad_scale=ctx.seed_value)
        ctx.grad = grad
        return losses

    @staticmethod
    def backward(ctx, g):
        if ctx.grad is None:
            return None, None, None, None, None
        return _scaled(ctx, g), None, None, None, None


def stage_frames_loss(label):
    """Start a per-frame loss whose three launches the caller spreads around its forward / backward pass
    (ops.CbceFramesStaged): counts the classes of ``label`` now.  None when the frames do not qualify (the one-call path
    then does everything).  Pass the result as ``staged=`` to class_balanced_cross_entropy_loss_frames - whose values are
    then written only by ``staged.finish()`` - and call that behind the backward pass."""
    if not label.is_cuda or label.dtype != torch.float32 or not label.is_contiguous() or label.dim() < 2:
        return None
    if (label.numel() // label.shape[0]) % 4 or label.data_ptr() % 16:
        return None
    return ops.CbceFramesStaged(label)


def class_balanced_cross_entropy_loss_frames(output, label, size_average=True, backward_seed=None, staged=None):
    """``class_balanced_cross_entropy_loss`` of each frame of a batch separately: a [N] tensor whose element i equals
    ``class_balanced_cross_entropy_loss(output[i:i+1], label[i:i+1], size_average)``.  Not in the reference; the
    online loop uses it when it runs several micro-batches of an accumulation cycle as one batched pass, where every
    frame must keep the class weights of its own [1,1,H,W] tensor (src/train_online.py:80 calls the loss per frame)."""
    if not output.is_cuda:
        raise RuntimeError("class_balanced_cross_entropy_loss_frames: the HIP implementation needs GPU tensors "
                           "(no CPU fallback)")
    if label.device != output.device:
        label = label.to(output.device)
    if tuple(label.shape) != tuple(output.shape):
        raise ValueError("class_balanced_cross_entropy_loss_frames: output {} vs label {}".format(
            tuple(output.shape), tuple(label.shape)))
    if staged is not None and (staged.label.data_ptr() != label.data_ptr() or tuple(staged.label.shape) != tuple(label.shape)):
        raise ValueError("class_balanced_cross_entropy_loss_frames: `staged` was started on another label tensor")
    return _CBCELossFrames.apply(output, label, size_average, backward_seed, staged)


def crop_offsets(size, target):
    """Pixels removed in front / behind by the reference's centre crop: floor(d/2) and ceil(d/2)."""
    d = size - target
    return -int(math.ceil(-d / 2.0)), -int(math.floor(-d / 2.0))


def center_crop(x, height, width):
    top, bottom = crop_offsets(int(x.size()[2]), height)
    left, right = crop_offsets(int(x.size()[3]), width)
    return F.pad(x, [-left, -right, -top, -bottom])


def upsample_filt(size):
    factor = (size + 1) // 2
    center = factor - 1 if size % 2 == 1 else factor - 0.5
    t = 1 - np.abs(np.arange(size) - center) / factor
    return t[:, None] * t[None, :]


def interp_surgery(lay):
    """Set a ConvTranspose2d so that it computes bilinear interpolation (diagonal, no groups)."""
    m, k, h, w = lay.weight.data.size()
    if m != k:
        raise Exception('input + output channels need to be the same')
    if h != w:
        raise Exception('filters need to be square')
    filt = torch.from_numpy(upsample_filt(h)).to(lay.weight.data.dtype)
    with torch.no_grad():
        lay.weight.data.zero_()
        idx = torch.arange(m)
        lay.weight.data[idx, idx] = filt.to(lay.weight.data.device)
    return lay.weight.data
