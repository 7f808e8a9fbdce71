"""SGD with momentum on one multi-tensor HIP launch per step (fosvos_sgd_momentum_step).

Same constructor and param-group semantics as ``torch.optim.SGD`` as the reference uses it
(src/util/network_provider.py:98-125,144-159: momentum 0.9, per-group lr / weight_decay, no
dampening, no nesterov).  It subclasses ``torch.optim.SGD`` so ``isinstance`` checks, ``param_groups``,
``zero_grad`` and ``state_dict`` behave as callers expect; only ``step`` is replaced.
"""
from __future__ import annotations

import ctypes

import torch
from torch import optim

from . import SgdEntry, check, lib


class FusedSGD(optim.SGD):
    def __init__(self, params, lr=1e-3, momentum=0, dampening=0, weight_decay=0, nesterov=False, **kw):
        if dampening != 0 or nesterov:
            raise NotImplementedError("FusedSGD implements the reference recipe: dampening=0, nesterov=False")
        kw.pop("foreach", None)
        kw.pop("fused", None)
        super().__init__(params, lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                         nesterov=nesterov, **kw)
        self._tables = {}  # tag -> the parameter table of that subset: entries, pinned staging copy, device copy, upload event

    @torch.no_grad()
    def step(self, closure=None, only=None, tag="all", zero_grad=False):
        """``only``: restrict this call to these parameters (an extension for loops that step parts of the model as soon as
        their gradients are final; every parameter must still be stepped exactly once per optimizer step by the caller).
        ``tag`` names the subset: the device-side table of a subset is kept under its tag and re-sent only when it changes.
        ``zero_grad``: the gradients of the stepped parameters are zeroed by the same kernel (``step(); zero_grad()`` of the
        reference's loops as one pass over the gradients; their ``.grad`` tensors stay allocated and hold zeros)."""
        only_ids = None if only is None else {id(p) for p in only}
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        entries = []
        momentum = None
        device = None
        max_numel = 0
        keep_alive = []
        zero_after = []
        updated = []
        for group in self.param_groups:
            if group.get("dampening", 0) != 0 or group.get("nesterov", False):
                raise NotImplementedError("FusedSGD: dampening/nesterov are not implemented")
            m = float(group["momentum"])
            if momentum is None:
                momentum = m
            elif m != momentum:
                raise NotImplementedError("FusedSGD: one momentum value for all groups (as in the reference recipe)")
            for p in group["params"]:
                if p.grad is None or (only_ids is not None and id(p) not in only_ids):
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedSGD: parameters must live on the GPU (no CPU fallback)")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise TypeError("FusedSGD: fp32 contiguous parameters only")
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.contiguous().float()
                    keep_alive.append(g)
                    if zero_grad:  # the kernel would zero the copy
                        zero_after.append(p.grad)
                st = self.state[p]
                buf = st.get("momentum_buffer")
                if buf is None:
                    # zero-initialised: momentum*0 + g == g, which is torch's "first step clones the gradient"
                    buf = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["momentum_buffer"] = buf
                device = p.device
                entries.append((p.data_ptr(), g.data_ptr(), buf.data_ptr(), p.numel(), float(group["lr"]),
                                float(group["weight_decay"])))
                if float(group["lr"]) != 0.0:  # lr 0 (the frozen upscale filters): the value cannot change
                    updated.append(p)
                max_numel = max(max_numel, p.numel())
        if not entries:
            return loss
        n = len(entries)
        nbytes = n * ctypes.sizeof(SgdEntry)
        # The device-side table only changes when a pointer, a size or a group's lr / weight decay does (the online loop:
        # never after its first step) - it is rebuilt and re-sent only then.  A rebuild rewrites the pinned staging
        # buffer, so it first waits for the previous upload to have been consumed.
        tab = self._tables.setdefault(tag, {"entries": None, "host": None, "dev": None, "event": None})
        if entries != tab["entries"]:
            if tab["host"] is None or tab["host"].numel() < nbytes:
                tab["host"] = torch.empty(max(nbytes, 4096), dtype=torch.uint8).pin_memory()
                tab["dev"] = torch.empty(max(nbytes, 4096), dtype=torch.uint8, device=device)
            if tab["event"] is not None:
                tab["event"].synchronize()
            arr = (SgdEntry * n).from_address(tab["host"].data_ptr())
            for i, e in enumerate(entries):
                arr[i].param, arr[i].grad, arr[i].momentum_buf, arr[i].numel, arr[i].lr, arr[i].weight_decay = e
            tab["dev"][:nbytes].copy_(tab["host"][:nbytes], non_blocking=True)
            tab["event"] = torch.cuda.Event()
            tab["event"].record()
            tab["entries"] = entries
        idx = device.index if device.index is not None else torch.cuda.current_device()
        check(lib().fosvos_sgd_momentum_step(tab["dev"].data_ptr(), n, max_numel, momentum, 2 if zero_grad else 0, idx,
                                             torch.cuda.current_stream(idx).cuda_stream), "sgd_momentum_step")
        # the kernel wrote through raw pointers: tell torch the tensors changed, so that version-keyed caches
        # (the packed bf16 weight images) and autograd's saved-tensor checks see the update
        for p in updated:
            torch.autograd.graph.increment_version(p)
        for g in zero_after:
            g.zero_()
        del keep_alive
        return loss
