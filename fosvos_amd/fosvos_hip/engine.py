"""Whole-network forward/backward of OSVOS-VGG on the HIP kernels, as ONE autograd node.

The reference builds this graph out of ~60 stock torch.nn calls (src/networks/osvos_vgg.py:61-83);
here the host issues the kernels directly (one Python frame per iteration, no per-layer autograd
bookkeeping) and keeps every intermediate in the layouts the kernels want:

  frame fp32 NCHW -> conv1_1 (fp32 VALU) -> bf16 NHWC activations -> MFMA convs / pools
  -> side_prep (fp32 NHWC, 16 ch) -> fused head -> fp32 [N,1,H,W] logits x 5

Backward walks the same chain in reverse: head -> per stage {side_prep wgrad+dgrad, conv wgrad+dgrad
(ReLU backward fused into the dgrad / pool-backward epilogues), pool backward} -> conv1_1 wgrad.
Weight masters stay fp32 (the nn.Parameters of the module); bf16 packed images are cached per
parameter version.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import os

import torch

from . import ops

STAGE_CHANNELS = ((64, 64), (128, 128), (256, 256, 256), (512, 512, 512), (512, 512, 512))
STAGE_IN = (3, 64, 128, 256, 512)
SIDE_CH = 16


def conv_module_index(stage: int, k: int) -> int:
    return 2 * k + (0 if stage == 0 else 1)


def param_names() -> List[str]:
    """state_dict order of the reference (src/networks/osvos_vgg.py:50-56)."""
    names = [f"upscale.{i}.weight" for i in range(4)] + [f"upscale_.{i}.weight" for i in range(4)]
    for s, chans in enumerate(STAGE_CHANNELS):
        for j in range(len(chans)):
            m = conv_module_index(s, j)
            names += [f"stages.{s}.{m}.weight", f"stages.{s}.{m}.bias"]
    for i in range(4):
        names += [f"side_prep.{i}.weight", f"side_prep.{i}.bias"]
    for i in range(4):
        names += [f"score_dsn.{i}.weight", f"score_dsn.{i}.bias"]
    names += ["fuse.weight", "fuse.bias"]
    return names


PARAM_NAMES = param_names()
PARAM_INDEX = {n: i for i, n in enumerate(PARAM_NAMES)}


class PackedWeights:
    """bf16 MFMA images of the 3x3 conv weights and the diagonal deconv filters, rebuilt only when
    the fp32 master changed (tracked by tensor version + storage pointer)."""

    def __init__(self) -> None:
        self._cache: Dict[str, Tuple[Tuple[int, int], object]] = {}
        self._uniform: Dict[str, bool] = {}  # deconv filter name -> its channel filters are identical (deconv_diag)
        self.arenas = ArenaPool()

    @staticmethod
    def _key(t: torch.Tensor) -> Tuple[int, int]:
        return (t.data_ptr(), t._version)

    def small(self, name: str, parts: Sequence[torch.Tensor]) -> torch.Tensor:
        """Concatenation of a few tiny parameters (the four score_dsn layers) as one contiguous vector."""
        key = tuple(self._key(t) for t in parts)
        hit = self._cache.get(name)
        if hit is not None and hit[0] == key:
            return hit[1]
        cat = torch.cat([t.detach().reshape(-1) for t in parts]).contiguous()
        self._cache[name] = (key, cat)
        return cat

    def conv(self, name: str, w: torch.Tensor):
        key = self._key(w)
        hit = self._cache.get(name)
        if hit is not None and hit[0] == key:
            return hit[1]
        packed = ops.pack_conv3x3_weights(w.detach(), True, True)
        self._cache[name] = (key, packed)
        return packed

    def conv_many(self, named: Sequence[Tuple[str, torch.Tensor]]):
        """conv() for a list of layers; all stale images are rebuilt by ONE launch."""
        stale = []
        for name, w in named:
            hit = self._cache.get(name)
            if hit is None or hit[0] != self._key(w):
                stale.append((name, w))
        if stale:
            packed = ops.pack_conv3x3_weights_multi([w.detach() for _, w in stale])
            for (name, w), pk in zip(stale, packed):
                self._cache[name] = (self._key(w), pk)
        return [self._cache[name][1] for name, _ in named]

    def deconv_diag(self, name: str, w: torch.Tensor) -> torch.Tensor:
        """[C,C,k,k] transposed-conv weight -> its diagonal, channel fastest: [k,k,C].  The head kernel applies one
        k x k filter per channel; the reference initialises these layers as diagonal bilinear filters and
        freezes them (src/layers/osvos_layers.py:70-81, lr 0 at src/util/network_provider.py:154-155).
        A weight with off-diagonal mass is refused loudly rather than silently mis-evaluated."""
        key = self._key(w)
        hit = self._cache.get(name)
        if hit is not None and hit[0] == key:
            return hit[1]
        wd = w.detach()
        c = wd.shape[0]
        idx = torch.arange(c, device=wd.device)
        diag = wd[idx, idx]  # [C,k,k]
        if c > 1:
            off = wd.abs().sum() - diag.abs().sum()
            if float(off) != 0.0:
                raise NotImplementedError(
                    f"{name}: transposed-conv weight has off-diagonal (cross-channel) entries; the HIP head "
                    f"implements the per-channel (diagonal) form the reference initialises and freezes")
        diag = diag.permute(1, 2, 0).contiguous()  # [k,k,C]
        self._cache[name] = (key, diag)
        # (the same sync as the off-diagonal check above, once per weight version): are the C channel filters identical?
        self._uniform[name] = bool((diag == diag[..., :1]).all().item())
        return diag

    def head_uniform_mask(self, P: Dict[str, torch.Tensor]) -> int:
        """Bit s: upscale[s]'s 16 channel filters are identical - interp_surgery's bilinear filters, which the optimizers
        never move (lr 0) - so the head kernels may contract the channels before the upsampling (fosvos_head_fwd's
        ``filt_uniform``).  Checked on the weights themselves whenever they change; FOSVOS_HEAD_UNIFORM=0 switches the
        fast path off (A/B)."""
        if os.environ.get("FOSVOS_HEAD_UNIFORM", "1") == "0":
            return 0
        mask = 0
        for i in range(4):
            name = f"upscale.{i}.weight"
            self.deconv_diag(name, P[name])
            if self._uniform.get(name, False):
                mask |= 1 << i
        return mask


def _conv_list():
    """(stage, k, cin, cout, weight name, bias name) for the 13 backbone convs."""
    out = []
    for s, chans in enumerate(STAGE_CHANNELS):
        cin = STAGE_IN[s]
        for j, cout in enumerate(chans):
            m = conv_module_index(s, j)
            out.append((s, j, cin, cout, f"stages.{s}.{m}.weight", f"stages.{s}.{m}.bias"))
            cin = cout
    return out


CONVS = _conv_list()


class Saved:
    __slots__ = ("frame", "conv_in", "conv_out", "pool_in", "feats", "side", "H", "W", "with_side_out", "filt", "filt1",
                 "filt_uniform")


def forward(P: Dict[str, torch.Tensor], packs: PackedWeights, x: torch.Tensor, with_side_out: bool = True,
            keep: bool = True):
    """Returns ([side_out x4 (or None), fused], Saved or None)."""
    if x.dim() != 4 or x.shape[1] != 3:
        raise ValueError(f"OSVOS_VGG expects [N,3,H,W] frames, got {tuple(x.shape)}")
    if not x.is_cuda:
        raise RuntimeError("the HIP OSVOS_VGG runs on the GPU only: move the module and the frame to cuda "
                           "(there is no CPU fallback)")
    x = x.contiguous().float()
    N, _, H, W = x.shape
    sv = Saved() if keep else None
    conv_in: List[torch.Tensor] = []
    conv_out: List[torch.Tensor] = []
    pool_in: List[Optional[torch.Tensor]] = []
    feats: List[torch.Tensor] = []
    side: List[torch.Tensor] = []
    a: Optional[torch.Tensor] = None
    ci_idx = 0
    for s, chans in enumerate(STAGE_CHANNELS):
        if s > 0:
            pool_in.append(a)
            a = ops.maxpool_fwd(a)
        for j in range(len(chans)):
            _, _, cin, cout, wn, bn = CONVS[ci_idx]
            ci_idx += 1
            if s == 0 and j == 0:
                y = ops.conv3x3_first_fwd(x, P[wn].detach(), P[bn].detach())
                conv_in.append(x)
            else:
                wf, _ = packs.conv(wn, P[wn])
                y = ops.conv3x3_fwd(a, wf, P[bn].detach(), cin, cout, relu=True)
                conv_in.append(a)
            conv_out.append(y)
            a = y
        feats.append(a)
        if s > 0:
            i = s - 1
            wf, _ = packs.conv(f"side_prep.{i}.weight", P[f"side_prep.{i}.weight"])
            side.append(ops.conv3x3_fwd(a, wf, P[f"side_prep.{i}.bias"].detach(), chans[-1], SIDE_CH, relu=False,
                                        out_f32=True))
    filt = [packs.deconv_diag(f"upscale.{i}.weight", P[f"upscale.{i}.weight"]) for i in range(4)]
    filt1 = [packs.deconv_diag(f"upscale_.{i}.weight", P[f"upscale_.{i}.weight"]).reshape(4 << i, 4 << i)
             for i in range(4)]
    dsn_w = torch.stack([P[f"score_dsn.{i}.weight"].detach().reshape(16) for i in range(4)]).contiguous()
    dsn_b = torch.cat([P[f"score_dsn.{i}.bias"].detach().reshape(1) for i in range(4)]).contiguous()
    fuse_w = P["fuse.weight"].detach().reshape(64).contiguous()
    fuse_b = P["fuse.bias"].detach().reshape(1).contiguous()
    umask = packs.head_uniform_mask(P)
    fused, side_out = ops.head_fwd(side, filt, filt1, dsn_w, dsn_b, fuse_w, fuse_b, H, W, with_side_out=with_side_out,
                                   filt_uniform=umask)
    if keep:
        sv.filt_uniform = umask
        sv.frame = x
        sv.conv_in, sv.conv_out, sv.pool_in, sv.feats, sv.side = conv_in, conv_out, pool_in, feats, side
        sv.H, sv.W, sv.with_side_out = H, W, with_side_out
        sv.filt, sv.filt1 = filt, filt1
    outs = (side_out if side_out is not None else [None] * 4) + [fused]
    return outs, sv


def backward(P: Dict[str, torch.Tensor], packs: PackedWeights, sv: "Saved", d_outs: Sequence[Optional[torch.Tensor]],
             inplace: bool = False) -> Dict[str, torch.Tensor]:
    """Gradients of sum_i <d_outs[i], out_i> wrt every parameter except the (frozen-by-recipe)
    transposed-conv weights.  d_outs[4] is the fused-logit gradient; d_outs[0..3] the side-output ones
    (all four or none).

    inplace=True (the training loops' gradient-accumulation mode): where a parameter already has a
    contiguous fp32 ``.grad``, the wgrad kernel ADDS into it directly and the parameter is left out of the
    returned dict - the accumulation ``p.grad += g`` that autograd would run as a separate pass per tensor
    becomes part of the kernel's epilogue."""
    grads: Dict[str, torch.Tensor] = {}

    def _ok(t: Optional[torch.Tensor]) -> bool:
        return t is not None and t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda

    def wgrad(wn: str, bn: str, x: torch.Tensor, dy: torch.Tensor, ci: int, co: int) -> None:
        pw, pb = P[wn], P[bn]
        if inplace and _ok(pw.grad) and _ok(pb.grad):
            ops.conv3x3_wgrad(x, dy, ci, co, dw=pw.grad, db=pb.grad, accumulate=True)
            torch.autograd.graph.increment_version(pw.grad)
            torch.autograd.graph.increment_version(pb.grad)
        else:
            grads[wn], grads[bn] = ops.conv3x3_wgrad(x, dy, ci, co)

    d_fused = d_outs[4]
    d_so = d_outs[:4]
    have_so = [g is not None for g in d_so]
    if any(have_so) and not all(have_so):
        ref = next(g for g in d_so if g is not None)
        d_so = [g if g is not None else torch.zeros_like(ref) for g in d_so]
    with_so = any(have_so)
    if d_fused is None and not with_so:
        return grads
    dsn_w = torch.stack([P[f"score_dsn.{i}.weight"].detach().reshape(16) for i in range(4)]).contiguous()
    fuse_w = P["fuse.weight"].detach().reshape(64).contiguous()
    d_side, d_fuse_w, d_fuse_b, d_dsn_w, d_dsn_b = ops.head_bwd(
        sv.side, sv.filt, sv.filt1 if with_so else None, dsn_w if with_so else None, fuse_w,
        d_fused.contiguous().float() if d_fused is not None else None,
        [g.contiguous().float() for g in d_so] if with_so else None, sv.H, sv.W,
        filt_uniform=getattr(sv, "filt_uniform", 0))
    grads["fuse.weight"] = d_fuse_w.reshape(1, 64, 1, 1)
    grads["fuse.bias"] = d_fuse_b
    if with_so:
        for i in range(4):
            grads[f"score_dsn.{i}.weight"] = d_dsn_w[i].reshape(1, 16, 1, 1)
            grads[f"score_dsn.{i}.bias"] = d_dsn_b[i:i + 1]

    g_pool: Optional[torch.Tensor] = None  # gradient wrt the current stage's output coming from the next stage's pool
    ci_idx = len(CONVS)
    for s in range(4, -1, -1):
        chans = STAGE_CHANNELS[s]
        feat = sv.feats[s]
        if s > 0:
            i = s - 1
            wn, bn = f"side_prep.{i}.weight", f"side_prep.{i}.bias"
            wgrad(wn, bn, feat, d_side[i], chans[-1], SIDE_CH)
            _, wd = packs.conv(wn, P[wn])
            # gradient wrt the stage output: ReLU-masked side dgrad + what came back through the next pool
            g = ops.conv3x3_dgrad(d_side[i], wd, chans[-1], SIDE_CH, relu_src=feat, addend=g_pool, out=g_pool)
        else:
            g = g_pool
        for j in range(len(chans) - 1, -1, -1):
            ci_idx -= 1
            _, _, cin, cout, wn, bn = CONVS[ci_idx]
            xin = sv.conv_in[ci_idx]
            if s == 0 and j == 0:
                dw, db = ops.conv3x3_first_wgrad(sv.frame, g)
                grads[wn], grads[bn] = dw, db
                break
            wgrad(wn, bn, xin, g, cin, cout)
            _, wd = packs.conv(wn, P[wn])
            # the conv input is a ReLU output when it came from a conv (mask here); when it came from a pool the
            # mask is applied by the pool backward below
            relu_src = xin if j > 0 else None
            g = ops.conv3x3_dgrad(g, wd, cin, cout, relu_src=relu_src)
        if s > 0:
            g_pool = ops.maxpool_bwd(sv.pool_in[s - 1], g, relu_mask=True)
    return grads


# ------------------------------------------------------------------------------------------------------
# Native layer loop (csrc/vgg_net.hip): one C-ABI call per forward, one per backward
# ------------------------------------------------------------------------------------------------------
import ctypes  # noqa: E402
import os  # noqa: E402

from . import Context, VggGrads, VggWeights, check, lib, ptr_array4  # noqa: E402

USE_NATIVE_LOOP = os.environ.get("FOSVOS_PY_ENGINE", "0") != "1"  # debugging switch: Python-driven per-op loop

_CONV_NAMES = [(wn, bn) for (_, _, _, _, wn, bn) in CONVS]


# The auxiliary streams of this process, one per device and role, created once and shared by every model: which hardware
# queue a HIP stream lands on depends on how many streams the process has created before it (ROCclr deals its - by default
# four - queues round-robin), and two streams on one hardware queue do not overlap at all.  A fresh stream per model (or per
# training call) therefore made the two-stream passes fast or slow by creation order (seen: 1290 vs 950 frames/s for the
# same loop, depending on what ran earlier in the process).  Fixed streams make the mapping the same for every model.
_SHARED_STREAMS: Dict[Tuple[int, str], "torch.cuda.Stream"] = {}
# What the probe below measured when a role's stream was created: {(device, role): {"overlaps_with": {name: bool}, "tries": n,
# "ms": {...}}}.  bench.py copies it into its JSON line (group1.pass_streams_overlap ...).
STREAM_PROBE: Dict[Tuple[int, str], dict] = {}
_PROBE_CYCLES = 400_000  # ~0.2 ms of spinning per kernel


def _streams_overlap(device_index: int, a: "torch.cuda.Stream", b: "torch.cuda.Stream") -> Tuple[bool, float, float]:
    """Do kernels on `a` and `b` run side by side?  Two spin kernels (torch.cuda._sleep), one per stream, timed on the host
    between device syncs against one spin kernel alone: two streams that share a hardware queue take twice as long as one.
    Returns (overlap, ms alone, ms of the pair).  Costs ~1 ms, once per pair and process."""
    import time
    dev = torch.device("cuda", device_index)

    def run(streams):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for st in streams:
            with torch.cuda.stream(st):
                torch.cuda._sleep(_PROBE_CYCLES)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) * 1e3

    run([a]); run([a, b])  # warm: first launches, clocks
    alone = min(run([a]) for _ in range(3))
    pair = min(run([a, b]) for _ in range(3))
    return pair < 1.5 * alone, alone, pair


def shared_stream(device_index: int, role: str) -> "torch.cuda.Stream":
    key = (device_index, role)
    st = _SHARED_STREAMS.get(key)
    if st is None:
        order = ("aux", "pass", "comm")  # creation order is part of the contract: the weight-gradient stream first
        for earlier in order[:order.index(role)] if role in order else ():
            shared_stream(device_index, earlier)
        # lab switch FOSVOS_STREAM_SKIP_<ROLE>=k: k throw-away streams first (shifts the role onto another hardware queue)
        skip = int(os.environ.get("FOSVOS_STREAM_SKIP_" + role.upper(), "0") or 0)
        _SHARED_STREAMS[("skipped", role, device_index)] = [torch.cuda.Stream(device=device_index) for _ in range(skip)]
        # The mapping is MEASURED, not trusted: the new stream must overlap with the caller's stream and with every role
        # created before it (the weight-gradient stream beside the data-gradient chain; the pass stream beside both).  If the
        # probe finds the pair serialised, the stream is parked (it keeps its queue slot) and the next one is tried.
        others = {"caller": torch.cuda.current_stream(device_index)}
        for r in order:
            if (device_index, r) in _SHARED_STREAMS:
                others[r] = _SHARED_STREAMS[(device_index, r)]
        probe_on = os.environ.get("FOSVOS_STREAM_PROBE", "1") != "0" and role != "comm" and hasattr(torch.cuda, "_sleep")
        parked, result = [], {"overlaps_with": {}, "tries": 0, "ms": {}}
        for attempt in range(6):
            st = torch.cuda.Stream(device=device_index)
            result["tries"] = attempt + 1
            if not probe_on:
                break
            ok = True
            for name, other in others.items():
                if other.cuda_stream == st.cuda_stream:
                    continue
                good, alone, pair = _streams_overlap(device_index, other, st)
                result["overlaps_with"][name] = bool(good)
                result["ms"][name] = [round(alone, 3), round(pair, 3)]
                ok = ok and good
            if ok:
                break
            parked.append(st)
        _SHARED_STREAMS[("parked", role, device_index)] = parked
        STREAM_PROBE[key] = result
        _SHARED_STREAMS[key] = st
    return st


class ArenaPool:
    """Activation/gradient/workspace arenas, one size per (N, H, W); the library never allocates."""

    def __init__(self) -> None:
        self._free: Dict[Tuple[int, int, int, int], List[torch.Tensor]] = {}
        self._aux: Dict[int, "torch.cuda.Stream"] = {}
        self._ctx: Dict[int, "Context"] = {}  # the model's execution context per device (events of its two-stream passes)
        # arenas (and the frames of their passes: conv1_1's weight gradient reads the frame itself) that deferred
        # weight-gradient kernels on the auxiliary stream may still read
        self._pending: List[Tuple[int, int, int, torch.Tensor, Optional[torch.Tensor]]] = []
        self._home: Dict[int, Tuple[int, int, int, int]] = {}  # arena address -> the key it was allocated for
        # a loop that batches up to this many frames of one size per pass (train_online: its group size) says so: a NEW arena
        # is then sized for every batch from the pass's own up to this one, so that a frame size is allocated once - not
        # again (hundreds of MB, milliseconds, in the middle of a cycle) the first time a larger group of it comes along
        self.reserve_frames = 0

    def aux_stream(self, device_index: int) -> int:
        """Handle of the auxiliary HIP stream the backward pass issues its weight-gradient kernels on
        (FOSVOS_TWO_STREAMS=0 disables it)."""
        if os.environ.get("FOSVOS_TWO_STREAMS", "1") == "0":
            return 0
        st = self._aux.get(device_index)
        if st is None:
            st = self._aux[device_index] = shared_stream(device_index, "aux")
        return st.cuda_stream

    def ctx(self, device_index: int):
        """Handle of this model's fosvos_ctx on `device_index` (created on first use, destroyed with the pool)."""
        c = self._ctx.get(device_index)
        if c is None:
            c = self._ctx[device_index] = Context(device_index)
        return c.handle

    def take(self, n: int, h: int, w: int, device: torch.device) -> torch.Tensor:
        dev = device.index if device.index is not None else torch.cuda.current_device()
        key = (n, h, w, dev)
        free = self._free.setdefault(key, [])
        if free:
            return free.pop()
        # a free arena of a LARGER batch at the same frame size serves as well (the layout is computed from the pass's own N
        # and only has to fit): a loop whose groups vary in size - frames of several shapes, bucketed per cycle - then
        # allocates per frame size, not per (group size, frame size), and no multi-gigabyte hipMalloc lands in a later cycle
        nbytes = lib().fosvos_vgg_arena_bytes(n, h, w)
        for (n2, h2, w2, d2), lst in self._free.items():
            # (an arena's size is not monotonic in N: fewer frames can mean more K splits and a larger workspace)
            if lst and h2 == h and w2 == w and d2 == dev and n2 > n and lst[-1].numel() >= nbytes + 256:
                return lst.pop()
        n_top = max(n, int(self.reserve_frames))
        for m in range(n + 1, n_top + 1):
            nbytes = max(nbytes, lib().fosvos_vgg_arena_bytes(m, h, w))
        arena = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
        self._home[arena.data_ptr()] = (n_top, h, w, dev)
        return arena

    def hold(self, n: int, h: int, w: int, arena: torch.Tensor, frame: Optional[torch.Tensor] = None) -> None:
        self._pending.append((n, h, w, arena, frame))

    def join(self) -> None:
        """Make the current stream wait for every deferred weight-gradient kernel, then recycle their arenas."""
        for idx, st in self._aux.items():
            torch.cuda.current_stream(idx).wait_stream(st)
        for n, h, w, arena, _frame in self._pending:
            self.give(n, h, w, arena)
        self._pending.clear()

    def give(self, n: int, h: int, w: int, arena: torch.Tensor) -> None:
        key = self._home.get(arena.data_ptr(), (n, h, w, arena.device.index))  # (back to the size it was allocated for)
        free = self._free.setdefault(key, [])
        if len(free) < 6:  # (a cycle of single-frame passes holds one arena per pass until its join)
            free.append(arena)
        else:
            self._home.pop(arena.data_ptr(), None)


def _aligned_ptr(arena: torch.Tensor) -> Tuple[int, int]:
    p = arena.data_ptr()
    ap = (p + 255) // 256 * 256
    return ap, arena.numel() - (ap - p)


def _weights_struct(P: Dict[str, torch.Tensor], packs: PackedWeights):
    """fosvos_vgg_weights for the current parameter values + the tensors it points into (kept alive by the caller)."""
    w = VggWeights()
    keep = []
    named = [(wn, P[wn]) for c, (wn, _) in enumerate(_CONV_NAMES) if c > 0]
    named += [(f"side_prep.{i}.weight", P[f"side_prep.{i}.weight"]) for i in range(4)]
    images = packs.conv_many(named)  # one launch repacks every layer an optimizer step touched
    for c, (wn, bn) in enumerate(_CONV_NAMES):
        w.conv_w[c] = P[wn].data_ptr()
        w.conv_b[c] = P[bn].data_ptr()
        if c > 0:
            wf, wd = images[c - 1]
            w.conv_wf[c], w.conv_wd[c] = wf.data_ptr(), wd.data_ptr()
    for i in range(4):
        wf, wd = images[len(_CONV_NAMES) - 1 + i]
        w.side_wf[i], w.side_wd[i] = wf.data_ptr(), wd.data_ptr()
        w.side_b[i] = P[f"side_prep.{i}.bias"].data_ptr()
        f = packs.deconv_diag(f"upscale.{i}.weight", P[f"upscale.{i}.weight"])
        f1 = packs.deconv_diag(f"upscale_.{i}.weight", P[f"upscale_.{i}.weight"])
        w.filt[i], w.filt1[i] = f.data_ptr(), f1.data_ptr()
    dsn_w = packs.small("dsn_w", [P[f"score_dsn.{i}.weight"] for i in range(4)])
    dsn_b = packs.small("dsn_b", [P[f"score_dsn.{i}.bias"] for i in range(4)])
    w.dsn_w, w.dsn_b = dsn_w.data_ptr(), dsn_b.data_ptr()
    w.fuse_w, w.fuse_b = P["fuse.weight"].data_ptr(), P["fuse.bias"].data_ptr()
    w.filt_uniform = packs.head_uniform_mask(P)
    keep += [dsn_w, dsn_b]
    return w, keep


def native_forward(P, packs, pool: ArenaPool, x: torch.Tensor, with_side_out: bool, keep: bool):
    if x.dim() != 4 or x.shape[1] != 3:
        raise ValueError(f"OSVOS_VGG expects [N,3,H,W] frames, got {tuple(x.shape)}")
    if not x.is_cuda:
        raise RuntimeError("the HIP OSVOS_VGG runs on the GPU only: move the module and the frame to cuda "
                           "(there is no CPU fallback)")
    for name in ("stages.0.0.weight", "fuse.weight"):
        if not P[name].is_cuda or P[name].dtype != torch.float32:
            raise RuntimeError("OSVOS_VGG parameters must be fp32 tensors on the GPU")
    x = x.contiguous().float()
    N, _, H, W = x.shape
    dev = x.device
    arena = pool.take(N, H, W, dev)
    ap, an = _aligned_ptr(arena)
    w, keep_alive = _weights_struct(P, packs)
    fused = torch.empty((N, 1, H, W), dtype=torch.float32, device=dev)
    outs = [torch.empty((N, 1, H, W), dtype=torch.float32, device=dev) for _ in range(4)] if with_side_out else None
    so = ptr_array4([o.data_ptr() for o in outs]) if with_side_out else None
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    t0 = ops._pb()
    # batched passes: the side_prep convs ride on the auxiliary stream beside the next stage's backbone convs (+0.4 % on the
    # five-frame training pass; FOSVOS_FWD_AUX=0: one stream).  A single frame stays on one stream: its kernels are too
    # short for the four event pairs to pay (inference protocol: 0.586 vs 0.564 ms per frame).
    # (forward_one_stream: the online loop sets it while it runs the passes of a cycle on two alternating streams - the
    # auxiliary stream is then busy with the previous pass's weight gradients, and the second chain would queue behind them)
    aux = (pool.aux_stream(idx) if N >= 2 and os.environ.get("FOSVOS_FWD_AUX", "1") != "0"
           and not getattr(packs, "forward_one_stream", False) else 0)
    if aux:
        check(lib().fosvos_vgg_forward_streams(pool.ctx(idx), ctypes.byref(w), x.data_ptr(), N, H, W, ap, an, fused.data_ptr(),
                                               so, torch.cuda.current_stream(idx).cuda_stream, aux), "vgg_forward_streams")
    else:
        check(lib().fosvos_vgg_forward(ctypes.byref(w), x.data_ptr(), N, H, W, ap, an, fused.data_ptr(), so, idx,
                                       torch.cuda.current_stream(idx).cuda_stream), "vgg_forward")
    ops._pe(t0, "vgg_forward", 2.0 * 129.114e9 * N * H * W / (480 * 854), 0.0)
    if not keep:
        pool.give(N, H, W, arena)
        arena = None
    return (outs if outs is not None else [None] * 4) + [fused], (arena, x, w, keep_alive, (N, H, W))


def native_backward(P, packs, saved, d_outs, inplace: bool, defer_join: bool = False) -> Dict[str, torch.Tensor]:
    arena, x, w, keep_alive, (N, H, W) = saved
    dev = x.device
    d_fused = d_outs[4]
    d_so = list(d_outs[:4])
    have = [g is not None for g in d_so]
    with_so = any(have)
    if with_so and not all(have):
        ref = next(g for g in d_so if g is not None)
        d_so = [g if g is not None else torch.zeros_like(ref) for g in d_so]
    grads: Dict[str, torch.Tensor] = {}
    if d_fused is None and not with_so:
        return grads
    g = VggGrads()
    hold = []

    def slot(name: str) -> int:
        p = P[name]
        if inplace:
            if p.grad is None or p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
                p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
            torch.autograd.graph.increment_version(p.grad)
            return p.grad.data_ptr()
        t = torch.empty_like(p, memory_format=torch.contiguous_format)
        grads[name] = t
        return t.data_ptr()

    for c, (wn, bn) in enumerate(_CONV_NAMES):
        g.conv_w[c], g.conv_b[c] = slot(wn), slot(bn)
    for i in range(4):
        g.side_w[i], g.side_b[i] = slot(f"side_prep.{i}.weight"), slot(f"side_prep.{i}.bias")
    g.fuse_w, g.fuse_b = slot("fuse.weight"), slot("fuse.bias")
    dsn_tmp = None
    if with_so:
        # the four score_dsn layers are separate [1,16,1,1] / [1] parameters: gather through one [4,16] / [4] scratch
        dsn_tmp = (torch.zeros((4, 16), dtype=torch.float32, device=dev), torch.zeros((4,), dtype=torch.float32, device=dev))
        g.dsn_w, g.dsn_b = dsn_tmp[0].data_ptr(), dsn_tmp[1].data_ptr()
    # overwrite_grads (set by a loop for the FIRST backward pass of an accumulation cycle, osvos_vgg.OSVOS_VGG.overwrite_grads):
    # the pass writes its gradients instead of adding them to what the buffers hold - the buffers then need no zeroing
    # between cycles (one write and one read of every gradient less per cycle).  Only the gradients this call produces are
    # written: the side-output layers' (score_dsn) go through a scratch and an add below, so the two do not combine.
    overwrite = inplace and getattr(packs, "overwrite_grads", False)
    if overwrite and with_so:
        raise RuntimeError("OSVOS_VGG backward: overwrite_grads with side-output gradients (score_dsn accumulates through a scratch)")
    g.accumulate = 1 if inplace and not overwrite else 0
    aux = packs.arenas.aux_stream(dev.index if dev.index is not None else torch.cuda.current_device())
    g.defer_join = 1 if (defer_join and inplace and aux) else 0
    g.bucket_events = 1 if getattr(packs, "publish_grad_buckets", False) else 0
    g.last_pass_of_cycle = 1 if getattr(packs, "last_pass_of_cycle", False) else 0
    if d_fused is not None:
        d_fused = d_fused.contiguous().float()
        hold.append(d_fused)
    dso = None
    if with_so:
        d_so = [t.contiguous().float() for t in d_so]
        hold += d_so
        dso = ptr_array4([t.data_ptr() for t in d_so])
    ap, an = _aligned_ptr(arena)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    # the dsn scratch is overwritten, never accumulated, by a separate flag-free path: run with accumulate for the
    # parameter buffers and add the scratch afterwards
    t0 = ops._pb()
    check(lib().fosvos_vgg_backward(packs.arenas.ctx(idx), ctypes.byref(w), ctypes.byref(g), x.data_ptr(), N, H, W, ap, an,
                                    d_fused.data_ptr() if d_fused is not None else None, dso,
                                    torch.cuda.current_stream(idx).cuda_stream, aux or None),
          "vgg_backward")
    ops._pe(t0, "vgg_backward", 2.0 * (2 * 129.114e9 - 0.708e9) * N * H * W / (480 * 854), 0.0)
    if with_so:
        for i in range(4):
            gw, gb = dsn_tmp[0][i].reshape(1, 16, 1, 1), dsn_tmp[1][i:i + 1]
            if inplace:
                for name, val in ((f"score_dsn.{i}.weight", gw), (f"score_dsn.{i}.bias", gb)):
                    p = P[name]
                    if p.grad is None:
                        p.grad = val.clone()
                    else:
                        p.grad.add_(val)
            else:
                grads[f"score_dsn.{i}.weight"], grads[f"score_dsn.{i}.bias"] = gw.clone(), gb.clone()
    del hold, keep_alive
    if g.defer_join:
        packs.arenas.hold(N, H, W, arena, x)
    else:
        packs.arenas.give(N, H, W, arena)
    return grads


class _OSVOSFunction(torch.autograd.Function):
    """One autograd node for the whole network.  Inputs: the frame and the 52 parameters in state_dict
    order; outputs: the 5 logit maps."""

    @staticmethod
    def forward(ctx, packs: PackedWeights, with_side_out: bool, inplace: bool, x: torch.Tensor, *params: torch.Tensor):
        P = dict(zip(PARAM_NAMES, params))
        if USE_NATIVE_LOOP:
            outs, sv = native_forward(P, packs, packs.arenas, x, with_side_out, keep=True)
        else:
            outs, sv = forward(P, packs, x, with_side_out=with_side_out, keep=True)
        ctx.sv = sv
        ctx.packs = packs
        ctx.P = P
        ctx.set_materialize_grads(False)
        ctx.with_side_out = with_side_out
        ctx.inplace = inplace
        if not with_side_out:
            outs = [torch.empty(0, device=x.device) for _ in range(4)] + [outs[4]]
            ctx.mark_non_differentiable(*outs[:4])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *d_outs):
        if ctx.sv is None:
            raise RuntimeError("OSVOS_VGG backward called but no activations were kept")
        d = list(d_outs)
        if not ctx.with_side_out:
            d = [None] * 4 + [d[4]]
        if USE_NATIVE_LOOP:
            grads = native_backward(ctx.P, ctx.packs, ctx.sv, d, inplace=ctx.inplace,
                                    defer_join=getattr(ctx.packs, "defer_wgrad_join", False))
        else:
            grads = backward(ctx.P, ctx.packs, ctx.sv, d, inplace=ctx.inplace)
        ctx.sv = None  # free the activations
        out = [None, None, None, None]
        for name in PARAM_NAMES:
            g = grads.get(name)
            p = ctx.P[name]
            if g is not None and tuple(g.shape) != tuple(p.shape):
                g = g.reshape(p.shape)
            out.append(g if p.requires_grad else None)
        return tuple(out)


def run(packs: PackedWeights, params: Sequence[torch.Tensor], x: torch.Tensor, with_side_out: bool = True,
        inplace_grad: bool = False):
    """Forward through the HIP kernels.  With grad mode on and trainable parameters this records one
    autograd node; otherwise (inference) nothing is kept."""
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        return list(_OSVOSFunction.apply(packs, with_side_out, inplace_grad, x, *params))
    if USE_NATIVE_LOOP:
        outs, _ = native_forward(dict(zip(PARAM_NAMES, params)), packs, packs.arenas, x, with_side_out, keep=False)
    else:
        outs, _ = forward(dict(zip(PARAM_NAMES, params)), packs, x, with_side_out=with_side_out, keep=False)
    if not with_side_out:
        outs = [torch.empty(0, device=x.device) for _ in range(4)] + [outs[4]]
    return outs
