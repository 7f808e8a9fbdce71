"""Tensor-level wrappers over the C ABI: check devices/dtypes/contiguity, pass raw pointers and the
current HIP stream, raise on any non-zero return.  No arithmetic happens in Python here.

Activations are torch.bfloat16 NHWC tensors ([N,H,W,C]); frames / logits are fp32 NCHW as in the
reference; side maps are fp32 NHWC [N,h,w,16].
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import CONV_OUT_F32, CONV_RELU, check, int_array4, lib, ptr_array4

_BF16 = torch.bfloat16
_F32 = torch.float32


def _need(t: torch.Tensor, dtype, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor must live on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{what}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{what}: tensor must be contiguous")
    return t


def _ctx(t: torch.Tensor) -> Tuple[int, int]:
    dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
    return dev, torch.cuda.current_stream(dev).cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _ru(a: int, b: int) -> int:
    return (a + b - 1) // b * b


class Workspace:
    """Grow-only scratch buffer per device AND stream (the library never allocates): ops on one stream are ordered and may
    share their scratch, two models driven on two streams (two host threads) must not."""

    def __init__(self) -> None:
        self._buf = {}

    def get(self, nbytes: int, device: torch.device) -> Tuple[Optional[int], int]:
        if nbytes <= 0:
            return None, 0
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = (idx, torch.cuda.current_stream(idx).cuda_stream)
        buf = self._buf.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(_ru(nbytes, 1 << 20), dtype=torch.uint8, device=device)
            self._buf[key] = buf
        return buf.data_ptr(), buf.numel()


_WS = Workspace()


# ------------------------------------------------------------------------------------------ profiling
class OpProfiler:
    """Optional per-call timing with HIP events on the launch stream (the current torch stream, which is
    the stream every kernel here is launched on).  Off by default: bench.py turns it on for a few
    iterations AFTER its timed region to obtain per-kernel durations for the roofline line."""

    def __init__(self, detail: bool = False) -> None:
        self.records = []  # (name, flops, bytes, start_event, end_event)
        self.detail = detail  # per-shape names for the generic conv (tests/bench_resnet_infer.py --detail)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, flops, nbytes, e0, e1 in self.records:
            a = agg.setdefault(name, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            a["calls"] += 1
            a["ms"] += e0.elapsed_time(e1)
            a["flops"] += flops
            a["bytes"] += nbytes
        return agg


_PROF: Optional[OpProfiler] = None


def set_profiler(p: Optional[OpProfiler]) -> None:
    global _PROF
    _PROF = p


def _pb():
    if _PROF is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _pe(e0, name: str, flops: float = 0.0, nbytes: float = 0.0) -> None:
    if e0 is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    _PROF.records.append((name, float(flops), float(nbytes), e0, e1))


# ------------------------------------------------------------------------------------------ layout
def nchw_to_nhwc_bf16(x: torch.Tensor, c_pad: Optional[int] = None) -> torch.Tensor:
    _need(x, _F32, "nchw_to_nhwc_bf16")
    n, c, h, w = x.shape
    c_pad = _ru(c, 8) if c_pad is None else c_pad
    out = torch.empty((n, h, w, c_pad), dtype=_BF16, device=x.device)
    dev, st = _ctx(x)
    check(lib().fosvos_nchw_f32_to_nhwc_bf16(x.data_ptr(), out.data_ptr(), n, c, h, w, c_pad, dev, st), "nchw_f32_to_nhwc_bf16")
    return out


def nhwc_bf16_to_nchw(x: torch.Tensor, c: Optional[int] = None) -> torch.Tensor:
    _need(x, _BF16, "nhwc_bf16_to_nchw")
    n, h, w, c_pad = x.shape
    c = c_pad if c is None else c
    out = torch.empty((n, c, h, w), dtype=_F32, device=x.device)
    dev, st = _ctx(x)
    check(lib().fosvos_nhwc_bf16_to_nchw_f32(x.data_ptr(), out.data_ptr(), n, c, h, w, c_pad, dev, st), "nhwc_bf16_to_nchw_f32")
    return out


def nhwc_f32_to_nchw(x: torch.Tensor) -> torch.Tensor:
    _need(x, _F32, "nhwc_f32_to_nchw")
    n, h, w, c = x.shape
    out = torch.empty((n, c, h, w), dtype=_F32, device=x.device)
    dev, st = _ctx(x)
    check(lib().fosvos_nhwc_f32_to_nchw_f32(x.data_ptr(), out.data_ptr(), n, c, h, w, dev, st), "nhwc_f32_to_nchw_f32")
    return out


def nchw_to_nhwc_f32(x: torch.Tensor) -> torch.Tensor:
    _need(x, _F32, "nchw_to_nhwc_f32")
    n, c, h, w = x.shape
    out = torch.empty((n, h, w, c), dtype=_F32, device=x.device)
    dev, st = _ctx(x)
    check(lib().fosvos_nchw_f32_to_nhwc_f32(x.data_ptr(), out.data_ptr(), n, c, h, w, dev, st), "nchw_f32_to_nhwc_f32")
    return out


# ------------------------------------------------------------------------------------------ weights
def pack_conv3x3_weights(w: torch.Tensor, want_fwd: bool = True, want_dgrad: bool = True):
    """fp32 OIHW -> (fwd image, dgrad image) as flat bf16 tensors (None when not wanted)."""
    _need(w, _F32, "pack_conv3x3_weights")
    co, ci, kh, kw = w.shape
    if (kh, kw) != (3, 3):
        raise ValueError("pack_conv3x3_weights: 3x3 kernels only")
    L = lib()
    fwd = torch.empty(L.fosvos_packed_weight_elems(co, ci), dtype=_BF16, device=w.device) if want_fwd else None
    dgr = torch.empty(L.fosvos_packed_weight_elems(ci, co), dtype=_BF16, device=w.device) if want_dgrad else None
    dev, st = _ctx(w)
    t0 = _pb()
    check(L.fosvos_pack_conv3x3_weights(w.data_ptr(), co, ci, _p(fwd), _p(dgr), dev, st), "pack_conv3x3_weights")
    _pe(t0, "pack_weights", 0.0, w.numel() * 4 + 2 * ((fwd.numel() if fwd is not None else 0) + (dgr.numel() if dgr is not None else 0)))
    return fwd, dgr


def pack_conv3x3_weights_multi(ws: Sequence[torch.Tensor]):
    """[(fwd image, dgrad image)] of several fp32 OIHW weights in ONE launch (in channels a multiple of 32)."""
    from . import PackEntry
    if not ws:
        return []
    L = lib()
    arr = (PackEntry * len(ws))()
    out = []
    nbytes = 0
    for i, w in enumerate(ws):
        _need(w, _F32, "pack_conv3x3_weights_multi")
        co, ci, kh, kw = w.shape
        if (kh, kw) != (3, 3) or w.device != ws[0].device:
            raise ValueError("pack_conv3x3_weights_multi: 3x3 kernels on one device only")
        fwd = torch.empty(L.fosvos_packed_weight_elems(co, ci), dtype=_BF16, device=w.device)
        dgr = torch.empty(L.fosvos_packed_weight_elems(ci, co), dtype=_BF16, device=w.device)
        arr[i].w, arr[i].w_fwd, arr[i].w_dgrad, arr[i].Co, arr[i].Ci = w.data_ptr(), fwd.data_ptr(), dgr.data_ptr(), co, ci
        out.append((fwd, dgr))
        nbytes += w.numel() * 4 + 2 * (fwd.numel() + dgr.numel())
    dev, st = _ctx(ws[0])
    t0 = _pb()
    check(L.fosvos_pack_conv3x3_weights_multi(arr, len(ws), dev, st), "pack_conv3x3_weights_multi")
    _pe(t0, "pack_weights", 0.0, nbytes)
    return out


# ------------------------------------------------------------------------------------------ conv
def conv3x3_first_fwd(frame: torch.Tensor, w: torch.Tensor, b: torch.Tensor, want_bits: bool = False):
    """y bf16 NHWC; want_bits: (y, relu_bits) with relu_bits uint8 [N,H,W,Co/8], bit e of byte g = (y[..., 8 g + e] > 0)
    (fosvos_conv3x3_first_fwd_bits: what conv3x3_dgrad takes as ``relu_bits``)."""
    _need(frame, _F32, "conv3x3_first_fwd frame"); _need(w, _F32, "conv3x3_first_fwd weight"); _need(b, _F32, "conv3x3_first_fwd bias")
    n, c, h, wd = frame.shape
    if c != 3 or tuple(w.shape[1:]) != (3, 3, 3):
        raise ValueError("conv3x3_first_fwd: expects a 3-channel frame and [Co,3,3,3] weights")
    co = w.shape[0]
    y = torch.empty((n, h, wd, co), dtype=_BF16, device=frame.device)
    dev, st = _ctx(frame)
    t0 = _pb()
    bits = torch.empty((n, h, wd, co // 8), dtype=torch.uint8, device=frame.device) if want_bits else None
    check(lib().fosvos_conv3x3_first_fwd_bits(frame.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), _p(bits), n, h, wd, co,
                                              dev, st), "conv3x3_first_fwd")
    _pe(t0, "conv1_1_fwd", 2.0 * n * h * wd * 27 * co, n * h * wd * (12 + 2 * co))
    return (y, bits) if want_bits else y


def conv3x3_first_wgrad(frame: torch.Tensor, dy: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    _need(frame, _F32, "conv3x3_first_wgrad frame"); _need(dy, _BF16, "conv3x3_first_wgrad dy")
    n, c, h, wd = frame.shape
    co = dy.shape[3]
    if tuple(dy.shape[:3]) != (n, h, wd):
        raise ValueError("conv3x3_first_wgrad: dy shape mismatch")
    L = lib()
    dw = torch.empty((co, 3, 3, 3), dtype=_F32, device=frame.device)
    db = torch.empty((co,), dtype=_F32, device=frame.device)
    ws, wsn = _WS.get(L.fosvos_conv3x3_first_wgrad_workspace_bytes(n, h, wd, co), frame.device)
    dev, st = _ctx(frame)
    t0 = _pb()
    check(L.fosvos_conv3x3_first_wgrad(frame.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), n, h, wd, co, ws, wsn,
                                       dev, st), "conv3x3_first_wgrad")
    _pe(t0, "conv1_1_wgrad", 2.0 * n * h * wd * 27 * co, n * h * wd * (12 + 2 * co))
    return dw, db


def conv3x3_plan(n: int, h: int, w: int, in_ch: int, out_ch: int) -> dict:
    """The igemm instantiation fosvos_conv3x3_dgrad (and the forward forms the persistent kernel does not take) launch for
    this shape (host arithmetic only): {"tile": (tile_h, tile_w, tile_co), "k_splits", "workgroups", "persistent"}."""
    import ctypes
    from . import Conv3x3PlanInfo
    info = Conv3x3PlanInfo()
    check(lib().fosvos_conv3x3_plan(n, h, w, in_ch, out_ch, ctypes.byref(info)), "conv3x3_plan")
    return {"tile": (info.tile_h, info.tile_w, info.tile_co), "k_splits": info.k_splits, "workgroups": info.workgroups,
            "persistent": bool(info.persistent)}


def conv3x3_fwd_plan(n: int, h: int, w: int, in_ch: int, out_ch: int, relu: bool = True) -> dict:
    """What fosvos_conv3x3_fwd / _fwd_pool launch for a bf16-output forward conv of this shape: the persistent eight-wave
    kernel k_conv3x3_pp ("persistent": True, 256 workgroups) where its tiles fill the chip, else the igemm instantiation."""
    import ctypes
    from . import Conv3x3PlanInfo
    info = Conv3x3PlanInfo()
    check(lib().fosvos_conv3x3_fwd_plan(n, h, w, in_ch, out_ch, CONV_RELU if relu else 0, ctypes.byref(info)), "conv3x3_fwd_plan")
    return {"tile": (info.tile_h, info.tile_w, info.tile_co), "k_splits": info.k_splits, "workgroups": info.workgroups,
            "persistent": bool(info.persistent)}


def conv3x3_first_plan(n: int, h: int, w: int) -> Tuple[int, int]:
    """(tiles, persistent workgroups) of fosvos_conv3x3_first_fwd."""
    import ctypes
    tiles, wgs = ctypes.c_int(), ctypes.c_int()
    check(lib().fosvos_conv3x3_first_plan(n, h, w, ctypes.byref(tiles), ctypes.byref(wgs)), "conv3x3_first_plan")
    return tiles.value, wgs.value


def conv3x3_fwd(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], ci: int, co: int,
                relu: bool = True, out_f32: bool = False) -> torch.Tensor:
    _need(x, _BF16, "conv3x3_fwd x"); _need(w_packed, _BF16, "conv3x3_fwd packed weight")
    n, h, wd, cx = x.shape
    if cx != _ru(ci, 32):
        raise ValueError(f"conv3x3_fwd: x has {cx} channels, expected {_ru(ci, 32)} (Ci={ci} padded to 32)")
    L = lib()
    if w_packed.numel() != L.fosvos_packed_weight_elems(co, ci):
        raise ValueError("conv3x3_fwd: packed weight size does not match (Co, Ci)")
    if bias is not None:
        _need(bias, _F32, "conv3x3_fwd bias")
        if bias.numel() != co:
            raise ValueError("conv3x3_fwd: bias size")
    y = torch.empty((n, h, wd, co), dtype=_F32 if out_f32 else _BF16, device=x.device)
    flags = (CONV_RELU if relu else 0) | (CONV_OUT_F32 if out_f32 else 0)
    ws, wsn = _WS.get(L.fosvos_conv3x3_workspace_bytes(n, h, wd, ci, co), x.device)
    dev, st = _ctx(x)
    t0 = _pb()
    check(L.fosvos_conv3x3_fwd(x.data_ptr(), w_packed.data_ptr(), _p(bias), y.data_ptr(), n, h, wd, ci, co, flags, ws, wsn,
                               dev, st), "conv3x3_fwd")
    _pe(t0, "conv3x3_fwd", 2.0 * n * h * wd * 9 * ci * co,
        n * h * wd * (2 * cx + (4 if out_f32 else 2) * co) + 2 * 9 * ci * co)
    return y


def conv3x3_fwd_pool(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], ci: int, co: int,
                     relu: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """conv3x3_fwd (bf16 output) and the 2x2 ceil-mode max pool of its output from one launch: (y, pooled y)."""
    _need(x, _BF16, "conv3x3_fwd_pool x"); _need(w_packed, _BF16, "conv3x3_fwd_pool packed weight")
    n, h, wd, cx = x.shape
    if cx != _ru(ci, 32):
        raise ValueError(f"conv3x3_fwd_pool: x has {cx} channels, expected {_ru(ci, 32)}")
    L = lib()
    if w_packed.numel() != L.fosvos_packed_weight_elems(co, ci):
        raise ValueError("conv3x3_fwd_pool: packed weight size does not match (Co, Ci)")
    if bias is not None:
        _need(bias, _F32, "conv3x3_fwd_pool bias")
    y = torch.empty((n, h, wd, co), dtype=_BF16, device=x.device)
    yp = torch.empty((n, (h + 1) // 2, (wd + 1) // 2, co), dtype=_BF16, device=x.device)
    ws, wsn = _WS.get(L.fosvos_conv3x3_workspace_bytes(n, h, wd, ci, co), x.device)
    dev, st = _ctx(x)
    t0 = _pb()
    check(L.fosvos_conv3x3_fwd_pool(x.data_ptr(), w_packed.data_ptr(), _p(bias), y.data_ptr(), yp.data_ptr(), n, h, wd, ci,
                                    co, CONV_RELU if relu else 0, ws, wsn, dev, st), "conv3x3_fwd_pool")
    _pe(t0, "conv3x3_fwd", 2.0 * n * h * wd * 9 * ci * co, n * h * wd * (2 * cx + 2 * co) + 2 * 9 * ci * co + yp.numel() * 2)
    return y, yp


def conv3x3_dgrad(dy: torch.Tensor, w_dgrad_packed: torch.Tensor, ci: int, co: int,
                  relu_src: Optional[torch.Tensor] = None, addend: Optional[torch.Tensor] = None,
                  out: Optional[torch.Tensor] = None, relu_bits: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx[N,H,W,Ci] = mask_{relu_src>0}(dgrad(dy)) + addend.  ``out`` may alias ``addend``.  ``relu_bits`` (uint8
    [N,H,W,Ci/8], instead of ``relu_src``): the same mask as one bit per element (fosvos_conv3x3_dgrad_bits)."""
    _need(dy, _BF16, "conv3x3_dgrad dy"); _need(w_dgrad_packed, _BF16, "conv3x3_dgrad packed weight")
    n, h, wd, cy = dy.shape
    if cy != _ru(co, 32):
        raise ValueError(f"conv3x3_dgrad: dy has {cy} channels, expected {_ru(co, 32)}")
    L = lib()
    if w_dgrad_packed.numel() != L.fosvos_packed_weight_elems(ci, co):
        raise ValueError("conv3x3_dgrad: packed weight size does not match (Ci, Co)")
    shape = (n, h, wd, ci)
    for t, nm in ((relu_src, "relu_src"), (addend, "addend"), (out, "out")):
        if t is not None:
            _need(t, _BF16, f"conv3x3_dgrad {nm}")
            if tuple(t.shape) != shape:
                raise ValueError(f"conv3x3_dgrad: {nm} shape {tuple(t.shape)} != {shape}")
    dx = out if out is not None else torch.empty(shape, dtype=_BF16, device=dy.device)
    ws, wsn = _WS.get(L.fosvos_conv3x3_workspace_bytes(n, h, wd, co, ci), dy.device)
    dev, st = _ctx(dy)
    t0 = _pb()
    if relu_bits is not None:
        if relu_src is not None or relu_bits.dtype != torch.uint8 or tuple(relu_bits.shape) != (n, h, wd, ci // 8) or \
                not relu_bits.is_contiguous() or not relu_bits.is_cuda:
            raise ValueError("conv3x3_dgrad: relu_bits must be a contiguous uint8 [N,H,W,Ci/8] GPU tensor, given instead of relu_src")
        check(L.fosvos_conv3x3_dgrad_bits(dy.data_ptr(), w_dgrad_packed.data_ptr(), relu_bits.data_ptr(), _p(addend),
                                          dx.data_ptr(), n, h, wd, ci, co, ws, wsn, dev, st), "conv3x3_dgrad_bits")
    else:
        check(L.fosvos_conv3x3_dgrad(dy.data_ptr(), w_dgrad_packed.data_ptr(), _p(relu_src), _p(addend), dx.data_ptr(), n, h, wd,
                                     ci, co, ws, wsn, dev, st), "conv3x3_dgrad")
    _pe(t0, "conv3x3_dgrad", 2.0 * n * h * wd * 9 * ci * co,
        n * h * wd * 2 * (cy + ci * (1 + (relu_src is not None) + (addend is not None))) + 2 * 9 * ci * co)
    return dx


def conv3x3_dgrad_unpool(dy: torch.Tensor, w_dgrad_packed: torch.Tensor, ci: int, co: int, x: torch.Tensor,
                         d_pooled: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx[N,H,W,Ci] = [x > 0] * dgrad(dy) + maxpool2x2_ceil_bwd(x, d_pooled): the data gradient of a conv whose input ``x`` (a
    post-ReLU stage output) also feeds the 2x2 ceil-mode max pool, the pool's backward in the same pass
    (fosvos_conv3x3_dgrad_unpool; bit for bit ``conv3x3_dgrad(..., relu_src=x, addend=maxpool2x2_ceil_bwd(x, d_pooled))``)."""
    _need(dy, _BF16, "conv3x3_dgrad_unpool dy"); _need(w_dgrad_packed, _BF16, "conv3x3_dgrad_unpool packed weight")
    _need(x, _BF16, "conv3x3_dgrad_unpool x"); _need(d_pooled, _BF16, "conv3x3_dgrad_unpool d_pooled")
    n, h, wd, cy = dy.shape
    if cy != _ru(co, 32):
        raise ValueError(f"conv3x3_dgrad_unpool: dy has {cy} channels, expected {_ru(co, 32)}")
    L = lib()
    if w_dgrad_packed.numel() != L.fosvos_packed_weight_elems(ci, co):
        raise ValueError("conv3x3_dgrad_unpool: packed weight size does not match (Ci, Co)")
    if tuple(x.shape) != (n, h, wd, ci) or tuple(d_pooled.shape) != (n, (h + 1) // 2, (wd + 1) // 2, ci):
        raise ValueError(f"conv3x3_dgrad_unpool: x {tuple(x.shape)} / d_pooled {tuple(d_pooled.shape)} do not match "
                         f"{(n, h, wd, ci)} and its ceil-mode pooled map")
    if out is not None:
        _need(out, _BF16, "conv3x3_dgrad_unpool out")
        if tuple(out.shape) != (n, h, wd, ci) or out.data_ptr() == x.data_ptr():
            raise ValueError("conv3x3_dgrad_unpool: out must have x's shape and must not alias it")
    dx = out if out is not None else torch.empty((n, h, wd, ci), dtype=_BF16, device=dy.device)
    ws, wsn = _WS.get(L.fosvos_conv3x3_workspace_bytes(n, h, wd, co, ci), dy.device)
    dev, st = _ctx(dy)
    t0 = _pb()
    check(L.fosvos_conv3x3_dgrad_unpool(dy.data_ptr(), w_dgrad_packed.data_ptr(), x.data_ptr(), d_pooled.data_ptr(), dx.data_ptr(),
                                        n, h, wd, ci, co, ws, wsn, dev, st), "conv3x3_dgrad_unpool")
    _pe(t0, "conv3x3_dgrad_unpool", 2.0 * n * h * wd * 9 * ci * co, n * h * wd * 2 * (cy + 2 * ci + ci // 4) + 2 * 9 * ci * co)
    return dx


def conv3x3_wgrad(x: torch.Tensor, dy: torch.Tensor, ci: int, co: int, with_bias: bool = True,
                  dw: Optional[torch.Tensor] = None, db: Optional[torch.Tensor] = None, accumulate: bool = False):
    _need(x, _BF16, "conv3x3_wgrad x"); _need(dy, _BF16, "conv3x3_wgrad dy")
    n, h, wd, cx = x.shape
    if cx != ci or tuple(dy.shape) != (n, h, wd, _ru(co, 32)):
        raise ValueError(f"conv3x3_wgrad: shapes x={tuple(x.shape)} dy={tuple(dy.shape)} do not match Ci={ci} Co={co}")
    L = lib()
    if dw is None:
        if accumulate:
            raise ValueError("conv3x3_wgrad: accumulate needs an existing dw")
        dw = torch.empty((co, ci, 3, 3), dtype=_F32, device=x.device)
    if with_bias and db is None:
        if accumulate:
            raise ValueError("conv3x3_wgrad: accumulate needs an existing db")
        db = torch.empty((co,), dtype=_F32, device=x.device)
    _need(dw, _F32, "conv3x3_wgrad dw")
    ws, wsn = _WS.get(L.fosvos_conv3x3_wgrad_workspace_bytes(n, h, wd, ci, co), x.device)
    dev, st = _ctx(x)
    # the two halves of fosvos_conv3x3_wgrad, timed apart: the MFMA kernel, then the slab reduction (memory-bound; the
    # network call batches the reductions of all layers into two launches)
    t0 = _pb()
    check(L.fosvos_conv3x3_wgrad_slabs(x.data_ptr(), dy.data_ptr(), 1 if with_bias else 0, n, h, wd, ci, co, ws, wsn, dev, st),
          "conv3x3_wgrad_slabs")
    _pe(t0, "conv3x3_wgrad", 2.0 * n * h * wd * 9 * ci * co, n * h * wd * 2 * (ci + dy.shape[3]))
    t0 = _pb()
    check(L.fosvos_conv3x3_wgrad_reduce(dw.data_ptr(), _p(db) if with_bias else None, n, h, wd, ci, co,
                                        1 if accumulate else 0, ws, wsn, dev, st), "conv3x3_wgrad_reduce")
    _pe(t0, "wgrad_reduce", 0.0, 4 * 9 * ci * co * 3)
    return dw, (db if with_bias else None)


def conv3x3_wgrad_one_call(x: torch.Tensor, dy: torch.Tensor, ci: int, co: int, with_bias: bool = True):
    """fosvos_conv3x3_wgrad: MFMA kernel and reduction behind one entry point (what a per-layer caller uses)."""
    _need(x, _BF16, "conv3x3_wgrad x"); _need(dy, _BF16, "conv3x3_wgrad dy")
    n, h, wd, _ = x.shape
    L = lib()
    dw = torch.empty((co, ci, 3, 3), dtype=_F32, device=x.device)
    db = torch.empty((co,), dtype=_F32, device=x.device) if with_bias else None
    ws, wsn = _WS.get(L.fosvos_conv3x3_wgrad_workspace_bytes(n, h, wd, ci, co), x.device)
    dev, st = _ctx(x)
    check(L.fosvos_conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _p(db), n, h, wd, ci, co, 0, ws, wsn, dev, st),
          "conv3x3_wgrad")
    return dw, db


# ------------------------------------------------------------------------------------------ pool
def maxpool_fwd(x: torch.Tensor) -> torch.Tensor:
    _need(x, _BF16, "maxpool_fwd")
    n, h, w, c = x.shape
    y = torch.empty((n, (h + 1) // 2, (w + 1) // 2, c), dtype=_BF16, device=x.device)
    dev, st = _ctx(x)
    t0 = _pb()
    check(lib().fosvos_maxpool2x2_ceil_fwd(x.data_ptr(), y.data_ptr(), n, h, w, c, dev, st), "maxpool2x2_ceil_fwd")
    _pe(t0, "maxpool_fwd", 0.0, 2 * (x.numel() + y.numel()))
    return y


def maxpool_bwd(x: torch.Tensor, dy: torch.Tensor, relu_mask: bool = True) -> torch.Tensor:
    _need(x, _BF16, "maxpool_bwd x"); _need(dy, _BF16, "maxpool_bwd dy")
    n, h, w, c = x.shape
    if tuple(dy.shape) != (n, (h + 1) // 2, (w + 1) // 2, c):
        raise ValueError("maxpool_bwd: dy shape mismatch")
    dx = torch.empty_like(x)
    dev, st = _ctx(x)
    t0 = _pb()
    check(lib().fosvos_maxpool2x2_ceil_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), n, h, w, c, 1 if relu_mask else 0,
                                           dev, st), "maxpool2x2_ceil_bwd")
    _pe(t0, "maxpool_bwd", 0.0, 2 * (2 * x.numel() + dy.numel()))
    return dx


# ------------------------------------------------------------------------------------------ head
def head_fwd(side: Sequence[torch.Tensor], filt: Sequence[torch.Tensor], filt1: Optional[Sequence[torch.Tensor]],
             dsn_w: Optional[torch.Tensor], dsn_b: Optional[torch.Tensor], fuse_w: torch.Tensor, fuse_b: torch.Tensor,
             H: int, W: int, with_side_out: bool = True, filt_uniform: int = 0):
    """side[s]: fp32 NHWC [N,hs,ws,16]; filt[s]: [k,k,16]; filt1[s]: [k,k]; dsn_w [4,16]; dsn_b [4];
    fuse_w [64]; fuse_b [1].  Returns (fused [N,1,H,W], [4 side outputs] or None).
    filt_uniform: bit s = the caller has checked that filt[s]'s 16 channel filters are identical (``filters_uniform_mask``)."""
    n = side[0].shape[0]
    for s in range(4):
        _need(side[s], _F32, "head_fwd side"); _need(filt[s], _F32, "head_fwd filt")
        k = 4 << s
        if side[s].shape[3] != 16 or tuple(filt[s].shape) != (k, k, 16):
            raise ValueError("head_fwd: side/filter shape")
    _need(fuse_w, _F32, "head_fwd fuse_w"); _need(fuse_b, _F32, "head_fwd fuse_b")
    dev_t = side[0].device
    fused = torch.empty((n, 1, H, W), dtype=_F32, device=dev_t)
    outs = None
    so_ptrs = [None] * 4
    f1_ptrs = [None] * 4
    if with_side_out:
        outs = [torch.empty((n, 1, H, W), dtype=_F32, device=dev_t) for _ in range(4)]
        so_ptrs = [o.data_ptr() for o in outs]
        for s in range(4):
            _need(filt1[s], _F32, "head_fwd filt1")
        f1_ptrs = [f.data_ptr() for f in filt1]
        _need(dsn_w, _F32, "head_fwd dsn_w"); _need(dsn_b, _F32, "head_fwd dsn_b")
    dev, st = _ctx(side[0])
    t0 = _pb()
    check(lib().fosvos_head_fwd(ptr_array4([t.data_ptr() for t in side]), int_array4([t.shape[1] for t in side]),
                                int_array4([t.shape[2] for t in side]), ptr_array4([t.data_ptr() for t in filt]),
                                ptr_array4(f1_ptrs), _p(dsn_w) if with_side_out else None,
                                _p(dsn_b) if with_side_out else None, fuse_w.data_ptr(), fuse_b.data_ptr(),
                                fused.data_ptr(), ptr_array4(so_ptrs), n, H, W, int(filt_uniform) & 15, dev, st), "head_fwd")
    _pe(t0, "head_fwd", 2.0 * n * H * W * 256, 4 * (sum(t.numel() for t in side) + n * H * W * (5 if with_side_out else 1)))
    return fused, outs


def filters_uniform_mask(filt: Sequence[torch.Tensor]) -> int:
    """Bit s set where filt[s] ([k,k,16], the diagonal of upscale[s].weight) holds the SAME k x k filter in all 16 channels,
    element for element - the promise head_fwd / head_bwd's ``filt_uniform`` asks for.  One device sync: call it when the
    weights change, not per step (engine.PackedWeights caches it with the filters)."""
    mask = 0
    for s, f in enumerate(filt):
        if bool((f == f[..., :1]).all().item()):
            mask |= 1 << s
    return mask


def head_bwd(side: Sequence[torch.Tensor], filt: Sequence[torch.Tensor], filt1: Optional[Sequence[torch.Tensor]],
             dsn_w: Optional[torch.Tensor], fuse_w: torch.Tensor, d_fused: Optional[torch.Tensor],
             d_side_out: Optional[Sequence[torch.Tensor]], H: int, W: int, filt_uniform: int = 0):
    """Returns (d_side[4] bf16 NHWC [N,hs,ws,32], d_fuse_w[64], d_fuse_b[1], d_dsn_w[4,16]|None, d_dsn_b[4]|None)."""
    n = side[0].shape[0]
    dev_t = side[0].device
    with_so = d_side_out is not None
    if d_fused is not None:
        _need(d_fused, _F32, "head_bwd d_fused")
    d_side = [torch.empty((n, t.shape[1], t.shape[2], 32), dtype=_BF16, device=dev_t) for t in side]
    d_fuse_w = torch.empty((64,), dtype=_F32, device=dev_t)
    d_fuse_b = torch.empty((1,), dtype=_F32, device=dev_t)
    d_dsn_w = torch.empty((4, 16), dtype=_F32, device=dev_t) if with_so else None
    d_dsn_b = torch.empty((4,), dtype=_F32, device=dev_t) if with_so else None
    dso_ptrs = [None] * 4
    f1_ptrs = [None] * 4
    if with_so:
        for s in range(4):
            _need(d_side_out[s], _F32, "head_bwd d_side_out"); _need(filt1[s], _F32, "head_bwd filt1")
        dso_ptrs = [t.data_ptr() for t in d_side_out]
        f1_ptrs = [t.data_ptr() for t in filt1]
        _need(dsn_w, _F32, "head_bwd dsn_w")
    L = lib()
    ws, wsn = _WS.get(L.fosvos_head_bwd_workspace_bytes(n, H, W), dev_t)
    dev, st = _ctx(side[0])
    t0 = _pb()
    check(L.fosvos_head_bwd(ptr_array4([t.data_ptr() for t in side]), int_array4([t.shape[1] for t in side]),
                            int_array4([t.shape[2] for t in side]), ptr_array4([t.data_ptr() for t in filt]),
                            ptr_array4(f1_ptrs), _p(dsn_w) if with_so else None, fuse_w.data_ptr(), _p(d_fused),
                            ptr_array4(dso_ptrs), ptr_array4([t.data_ptr() for t in d_side]), d_fuse_w.data_ptr(),
                            d_fuse_b.data_ptr(), _p(d_dsn_w), _p(d_dsn_b), n, H, W, int(filt_uniform) & 15, ws, wsn, dev, st),
          "head_bwd")
    _pe(t0, "head_bwd", 2.0 * n * H * W * 256, 4 * (sum(t.numel() for t in side) + n * H * W * (5 if with_so else 1)) +
        2 * sum(t.numel() for t in d_side))
    return d_side, d_fuse_w, d_fuse_b, d_dsn_w, d_dsn_b


# ------------------------------------------------------------------------------------------ loss
def cbce_loss(logits: torch.Tensor, label: torch.Tensor, size_average: bool = True, grad_scale: float = 1.0,
              want_grad: bool = True, batch_counts: Optional[torch.Tensor] = None):
    """Returns (loss: 0-dim fp32 tensor on the device, grad like logits or None).  batch_counts: device float64 [2]
    {positives, pixels} of the whole (data-parallel) batch this tensor is a shard of; None = count this tensor."""
    _need(logits, _F32, "cbce_loss logits"); _need(label, _F32, "cbce_loss label")
    if batch_counts is not None:
        if (not batch_counts.is_cuda or batch_counts.dtype != torch.float64 or batch_counts.numel() != 2
                or not batch_counts.is_contiguous()):
            raise ValueError("cbce_loss: batch_counts must be a contiguous float64 [2] tensor on the GPU")
    if logits.shape != label.shape:
        raise ValueError(f"cbce_loss: logits {tuple(logits.shape)} vs label {tuple(label.shape)}")
    # the kernels read float4: a slice of a batch (one rank's shard, one frame) may start off a 16-byte boundary
    if logits.data_ptr() % 16:
        logits = logits.clone()
    if label.data_ptr() % 16:
        label = label.clone()
    L = lib()
    loss = torch.empty((), dtype=_F32, device=logits.device)
    grad = torch.empty_like(logits) if want_grad else None
    # the loss keeps its own small workspace: it must stay valid until the kernels ran, and the shared
    # grow-only buffer may be re-used by the next op on the same stream (which is ordered after us)
    ws, wsn = _WS.get(L.fosvos_cbce_workspace_bytes(logits.numel()), logits.device)
    dev, st = _ctx(logits)
    t0 = _pb()
    if batch_counts is None:
        check(L.fosvos_cbce_loss(logits.data_ptr(), label.data_ptr(), logits.numel(), 1 if size_average else 0,
                                 float(grad_scale), loss.data_ptr(), _p(grad), ws, wsn, dev, st), "cbce_loss")
    else:
        check(L.fosvos_cbce_loss_batch_counts(logits.data_ptr(), label.data_ptr(), logits.numel(),
                                              1 if size_average else 0, float(grad_scale), batch_counts.data_ptr(),
                                              loss.data_ptr(), _p(grad), ws, wsn, dev, st), "cbce_loss_batch_counts")
    _pe(t0, "cbce_loss", 0.0, logits.numel() * (16 if want_grad else 12))
    return loss, grad


def cbce_loss_frames(logits: torch.Tensor, label: torch.Tensor, size_average: bool = True, want_grad: bool = True,
                     grad_scale: float = 1.0):
    """The loss of every frame of [N,1,H,W] logits on its own: ([N] losses, grad like logits or None), one set of launches.
    Frames whose element count is not a multiple of 4 go through one call per frame (16-byte alignment of the kernels)."""
    _need(logits, _F32, "cbce_loss_frames logits"); _need(label, _F32, "cbce_loss_frames label")
    if logits.shape != label.shape or logits.dim() < 2:
        raise ValueError(f"cbce_loss_frames: logits {tuple(logits.shape)} vs label {tuple(label.shape)}")
    n = logits.shape[0]
    per = logits.numel() // n
    if per % 4 or logits.data_ptr() % 16 or label.data_ptr() % 16:
        parts = [cbce_loss(logits[i:i + 1], label[i:i + 1], size_average=size_average, grad_scale=grad_scale,
                           want_grad=want_grad) for i in range(n)]
        return torch.stack([p[0] for p in parts]), (torch.cat([p[1] for p in parts]) if want_grad else None)
    L = lib()
    losses = torch.empty((n,), dtype=_F32, device=logits.device)
    grad = torch.empty_like(logits) if want_grad else None
    ws, wsn = _WS.get(n * L.fosvos_cbce_workspace_bytes(per), logits.device)
    dev, st = _ctx(logits)
    t0 = _pb()
    check(L.fosvos_cbce_loss_frames(logits.data_ptr(), label.data_ptr(), per, n, 1 if size_average else 0, float(grad_scale),
                                    losses.data_ptr(), _p(grad), ws, wsn, dev, st), "cbce_loss_frames")
    _pe(t0, "cbce_loss", 0.0, logits.numel() * (16 if want_grad else 12))
    return losses, grad


CBCE_COUNT, CBCE_LOSS, CBCE_FINISH = 1, 2, 4


class CbceFramesStaged(object):
    """``cbce_loss_frames`` in its three launches (fosvos_cbce_loss_frames_parts), for a loop that has a forward pass to
    put behind the first and a backward pass behind the second: the class counts need only the labels, the backward pass
    only the gradient - counting beside the forward pass and writing the loss VALUES behind the backward pass takes two
    small dependent launches (and the copy of the values to the host) off the path between the two passes.

        staged = CbceFramesStaged(label)              # counts; before the forward pass
        losses, grad = staged.loss(logits, ...)       # `losses` is allocated, NOT yet written
        ...backward pass...
        staged.finish()                               # now `losses` holds the values (stream order)

    Same stream for all three calls.  Frames whose element count is not a multiple of 4 are not supported here (the caller
    falls back to cbce_loss_frames)."""

    def __init__(self, label: torch.Tensor):
        _need(label, _F32, "CbceFramesStaged label")
        if label.dim() < 2:
            raise ValueError(f"CbceFramesStaged: label {tuple(label.shape)}")
        self.n = label.shape[0]
        self.per = label.numel() // self.n
        if self.per % 4 or label.data_ptr() % 16:
            raise ValueError("CbceFramesStaged: frames must be a multiple of 4 elements and 16-byte aligned")
        L = lib()
        self.label = label
        # the stages share this workspace across the passes in between: its own tensor, not the ops' scratch
        self.wsn = self.n * L.fosvos_cbce_workspace_bytes(self.per)
        self.ws = torch.empty((self.wsn,), dtype=torch.uint8, device=label.device)
        self.losses = None
        self.size_average = None
        dev, st = _ctx(label)
        t0 = _pb()
        check(L.fosvos_cbce_loss_frames_parts(None, label.data_ptr(), self.per, self.n, 0, 1.0, None, None, self.ws.data_ptr(),
                                              self.wsn, CBCE_COUNT, dev, st), "cbce_loss_frames_parts(count)")
        _pe(t0, "cbce_loss", 0.0, label.numel() * 4)

    def loss(self, logits: torch.Tensor, size_average: bool = True, want_grad: bool = True, grad_scale: float = 1.0):
        _need(logits, _F32, "CbceFramesStaged logits")
        if logits.shape != self.label.shape or logits.data_ptr() % 16:
            raise ValueError(f"CbceFramesStaged: logits {tuple(logits.shape)} vs label {tuple(self.label.shape)}")
        self.losses = torch.empty((self.n,), dtype=_F32, device=logits.device)
        self.size_average = bool(size_average)
        grad = torch.empty_like(logits) if want_grad else None
        dev, st = _ctx(logits)
        t0 = _pb()
        check(lib().fosvos_cbce_loss_frames_parts(logits.data_ptr(), self.label.data_ptr(), self.per, self.n,
                                                  1 if size_average else 0, float(grad_scale), None, _p(grad),
                                                  self.ws.data_ptr(), self.wsn, CBCE_LOSS, dev, st),
              "cbce_loss_frames_parts(loss)")
        _pe(t0, "cbce_loss", 0.0, logits.numel() * (12 if want_grad else 8))
        return self.losses, grad

    def finish(self) -> torch.Tensor:
        if self.losses is None:
            raise RuntimeError("CbceFramesStaged.finish before loss")
        dev, st = _ctx(self.losses)
        check(lib().fosvos_cbce_loss_frames_parts(None, None, self.per, self.n, 1 if self.size_average else 0, 1.0,
                                                  self.losses.data_ptr(), None, self.ws.data_ptr(), self.wsn, CBCE_FINISH,
                                                  dev, st), "cbce_loss_frames_parts(finish)")
        return self.losses


# ------------------------------------------------------------------------------------------ thin-channel ResNet path
def fold_conv_bn(w: torch.Tensor, conv_bias: Optional[torch.Tensor] = None, bn: Optional[Sequence] = None):
    """(w * s, bn_bias - mean * s [+ conv_bias * s]) with s = bn_weight / sqrt(var + eps): fp32 OIHW in and out."""
    _need(w, _F32, "fold_conv_bn weight")
    co, ci, k, k2 = w.shape
    if k != k2:
        raise ValueError("fold_conv_bn: square kernels only")
    out = torch.empty_like(w)
    bias = torch.empty((co,), dtype=_F32, device=w.device)
    bnp = [None] * 4
    eps = 0.0
    if bn is not None:
        for t in bn[:4]:
            _need(t, _F32, "fold_conv_bn BatchNorm tensor")
            if t.numel() != co:
                raise ValueError("fold_conv_bn: BatchNorm size != out channels")
        bnp = [t.data_ptr() for t in bn[:4]]
        eps = float(bn[4])
    if conv_bias is not None:
        _need(conv_bias, _F32, "fold_conv_bn conv bias")
    dev, st = _ctx(w)
    check(lib().fosvos_fold_conv_bn(w.data_ptr(), co, ci, k, _p(conv_bias), bnp[0], bnp[1], bnp[2], bnp[3], eps,
                                    out.data_ptr(), bias.data_ptr(), dev, st), "fold_conv_bn")
    return out, bias


def conv3x3_fwd_add(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, ci: int, co: int, relu: bool = True,
                    addend: Optional[torch.Tensor] = None, out_f32: bool = False) -> torch.Tensor:
    """MFMA conv3x3 in its residual form: act(conv(x) + bias + addend), ReLU after the add; bf16 NHWC in and out."""
    _need(x, _BF16, "conv3x3_fwd_add x"); _need(w_packed, _BF16, "conv3x3_fwd_add packed weight")
    _need(bias, _F32, "conv3x3_fwd_add bias")
    n, h, wd, cx = x.shape
    L = lib()
    if cx != ci or ci % 32 or (co % 64 and co != 32 and not (co == 16 and addend is None)):
        raise ValueError(f"conv3x3_fwd_add: x has {cx} channels for Ci={ci}, Co={co} (Ci % 32 == 0, Co = 32 or Co % 64 == 0)")
    if out_f32 and co != 16:
        raise ValueError("conv3x3_fwd_add: fp32 output is the 16-channel side_prep form")
    if w_packed.numel() != L.fosvos_packed_weight_elems(co, ci) or bias.numel() != co:
        raise ValueError("conv3x3_fwd_add: packed weight / bias size does not match (Co, Ci)")
    y = torch.empty((n, h, wd, co), dtype=_F32 if out_f32 else _BF16, device=x.device)
    if addend is not None:
        _need(addend, _BF16, "conv3x3_fwd_add addend")
        if addend.shape != y.shape:
            raise ValueError(f"conv3x3_fwd_add: addend {tuple(addend.shape)} vs output {tuple(y.shape)}")
    ws, wsn = _WS.get(L.fosvos_conv3x3_workspace_bytes(n, h, wd, ci, co), x.device)
    dev, st = _ctx(x)
    t0 = _pb()
    check(L.fosvos_conv3x3_fwd_add(x.data_ptr(), w_packed.data_ptr(), bias.data_ptr(), _p(addend), y.data_ptr(), n, h, wd, ci,
                                   co, (CONV_RELU if relu else 0) | (CONV_OUT_F32 if out_f32 else 0), ws, wsn, dev, st),
          "conv3x3_fwd_add")
    _pe(t0, f"mfma3x3s1 {ci}->{co} @{h}x{wd}" if _PROF is not None and _PROF.detail else "mfma3x3s1",
        2.0 * n * h * wd * 9 * ci * co, 2 * (x.numel() + y.numel()) + (2 * addend.numel() if addend is not None else 0))
    return y


def conv3x3_s2_fwd(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, ci: int, co: int,
                   relu: bool = True) -> torch.Tensor:
    """Stride-2 conv3x3 (pad 1) on the MFMA path: [N,H,W,ci] -> [N,(H-1)//2+1,(W-1)//2+1,co], bf16 NHWC."""
    _need(x, _BF16, "conv3x3_s2_fwd x"); _need(w_packed, _BF16, "conv3x3_s2_fwd packed weight")
    _need(bias, _F32, "conv3x3_s2_fwd bias")
    n, h, wd, cx = x.shape
    L = lib()
    if cx != ci or ci % 32 or (co % 64 and co != 32):
        raise ValueError(f"conv3x3_s2_fwd: x has {cx} channels for Ci={ci}, Co={co} (Ci % 32 == 0, Co = 32 or Co % 64 == 0)")
    if w_packed.numel() != L.fosvos_packed_weight_elems(co, ci) or bias.numel() != co:
        raise ValueError("conv3x3_s2_fwd: packed weight / bias size does not match (Co, Ci)")
    ho, wo = (h - 1) // 2 + 1, (wd - 1) // 2 + 1
    y = torch.empty((n, ho, wo, co), dtype=_BF16, device=x.device)
    ws, wsn = _WS.get(L.fosvos_conv3x3_workspace_bytes(n, h, wd, ci, co), x.device)
    dev, st = _ctx(x)
    t0 = _pb()
    check(L.fosvos_conv3x3_s2_fwd(x.data_ptr(), w_packed.data_ptr(), bias.data_ptr(), y.data_ptr(), n, h, wd, ci, co,
                                  CONV_RELU if relu else 0, ws, wsn, dev, st), "conv3x3_s2_fwd")
    _pe(t0, f"mfma3x3s2 {ci}->{co} @{ho}x{wo}" if _PROF is not None and _PROF.detail else "mfma3x3s2",
        2.0 * n * ho * wo * 9 * ci * co, 2 * (x.numel() + y.numel()))
    return y


def conv_out_size(h: int, k: int, stride: int) -> int:
    return (h + 2 * (k // 2) - k) // stride + 1


def pack_conv2d_bn(w: torch.Tensor, conv_bias: Optional[torch.Tensor] = None, bn: Optional[Sequence] = None):
    """w: fp32 OIHW [Co,Ci,k,k] (k = 1 or 3); bn = (weight, bias, running_mean, running_var, eps) of the eval-mode
    BatchNorm2d that follows the conv, or None.  Returns (packed int32 words, folded fp32 bias)."""
    _need(w, _F32, "pack_conv2d_bn weight")
    co, ci, k, k2 = w.shape
    if k != k2:
        raise ValueError("pack_conv2d_bn: square kernels only")
    L = lib()
    packed = torch.empty((L.fosvos_conv2d_packed_dwords(co, ci, k),), dtype=torch.int32, device=w.device)
    bias = torch.empty((L.fosvos_conv2d_bias_elems(co),), dtype=_F32, device=w.device)
    bnp = [None] * 4
    eps = 0.0
    if bn is not None:
        for t in bn[:4]:
            _need(t, _F32, "pack_conv2d_bn BatchNorm tensor")
            if t.numel() != co:
                raise ValueError("pack_conv2d_bn: BatchNorm size != out channels")
        bnp = [t.data_ptr() for t in bn[:4]]
        eps = float(bn[4])
    if conv_bias is not None:
        _need(conv_bias, _F32, "pack_conv2d_bn conv bias")
    dev, st = _ctx(w)
    check(L.fosvos_pack_conv2d_bn(w.data_ptr(), co, ci, k, _p(conv_bias), bnp[0], bnp[1], bnp[2], bnp[3], eps,
                                  packed.data_ptr(), bias.data_ptr(), dev, st), "pack_conv2d_bn")
    return packed, bias


def conv2d_fwd(x: torch.Tensor, packed: torch.Tensor, bias: torch.Tensor, ci: int, co: int, k: int, stride: int = 1,
               relu: bool = False, addend: Optional[torch.Tensor] = None, out_f32: bool = False) -> torch.Tensor:
    """x: bf16 NHWC [N,H,W,ru8(ci)] -> bf16 (or fp32) NHWC [N,Ho,Wo,ru8(co)] = act(conv + bias + addend)."""
    _need(x, _BF16, "conv2d_fwd x")
    _need(packed, torch.int32, "conv2d_fwd packed weights"); _need(bias, _F32, "conv2d_fwd bias")
    n, h, w, cp = x.shape
    L = lib()
    if cp != _ru(ci, 8):
        raise ValueError(f"conv2d_fwd: x has {cp} channels, expected {_ru(ci, 8)} (ci={ci} padded to 8)")
    if packed.numel() != L.fosvos_conv2d_packed_dwords(co, ci, k) or bias.numel() != L.fosvos_conv2d_bias_elems(co):
        raise ValueError("conv2d_fwd: packed image / bias size does not match (co, ci, k)")
    ho, wo = conv_out_size(h, k, stride), conv_out_size(w, k, stride)
    y = torch.empty((n, ho, wo, _ru(co, 8)), dtype=_F32 if out_f32 else _BF16, device=x.device)
    if addend is not None:
        _need(addend, _BF16, "conv2d_fwd addend")
        if addend.shape != y.shape:
            raise ValueError(f"conv2d_fwd: addend {tuple(addend.shape)} vs output {tuple(y.shape)}")
    flags = (CONV_RELU if relu else 0) | (CONV_OUT_F32 if out_f32 else 0)
    dev, st = _ctx(x)
    t0 = _pb()
    check(L.fosvos_conv2d_fwd(x.data_ptr(), packed.data_ptr(), bias.data_ptr(), _p(addend), y.data_ptr(), n, h, w, ci, co,
                              k, stride, flags, dev, st), "conv2d_fwd")
    _pe(t0, f"conv{k}x{k}s{stride} {ci}->{co} @{ho}x{wo}" if _PROF is not None and _PROF.detail else f"conv{k}x{k}s{stride}",
        2.0 * n * ho * wo * k * k * ci * co,
        2 * x.numel() + y.numel() * y.element_size() + (2 * addend.numel() if addend is not None else 0))
    return y


def pack_conv7x7_bn(w: torch.Tensor, bn: Optional[Sequence] = None):
    _need(w, _F32, "pack_conv7x7_bn weight")
    co, ci, k, k2 = w.shape
    if (ci, k, k2) != (3, 7, 7):
        raise ValueError(f"pack_conv7x7_bn: expected [Co,3,7,7], got {tuple(w.shape)}")
    L = lib()
    packed = torch.empty((L.fosvos_conv7x7_packed_elems(co),), dtype=_F32, device=w.device)
    bias = torch.empty((L.fosvos_conv2d_bias_elems(co),), dtype=_F32, device=w.device)
    bnp = [None] * 4
    eps = 0.0
    if bn is not None:
        for t in bn[:4]:
            _need(t, _F32, "pack_conv7x7_bn BatchNorm tensor")
        bnp = [t.data_ptr() for t in bn[:4]]
        eps = float(bn[4])
    dev, st = _ctx(w)
    check(L.fosvos_pack_conv7x7_bn(w.data_ptr(), co, bnp[0], bnp[1], bnp[2], bnp[3], eps, packed.data_ptr(),
                                   bias.data_ptr(), dev, st), "pack_conv7x7_bn")
    return packed, bias


CONV_FP32_MATH = 4


def conv7x7s2_first_fwd(frame: torch.Tensor, packed: torch.Tensor, bias: torch.Tensor, co: int,
                        relu: bool = True, fp32_math: bool = False) -> torch.Tensor:
    """frame: fp32 NCHW [N,3,H,W] -> bf16 NHWC [N,(H-1)//2+1,(W-1)//2+1,ru8(co)].  Default: bf16 MFMA (frame and
    weights rounded to bf16, fp32 accumulate); fp32_math keeps both fp32 on the vector ALU."""
    _need(frame, _F32, "conv7x7s2_first_fwd frame")
    _need(packed, _F32, "conv7x7s2_first_fwd packed weights"); _need(bias, _F32, "conv7x7s2_first_fwd bias")
    n, c, h, w = frame.shape
    L = lib()
    if c != 3:
        raise ValueError("conv7x7s2_first_fwd: 3-channel frames only")
    if packed.numel() != L.fosvos_conv7x7_packed_elems(co) or bias.numel() != L.fosvos_conv2d_bias_elems(co):
        raise ValueError("conv7x7s2_first_fwd: packed image / bias size does not match co")
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty((n, ho, wo, _ru(co, 8)), dtype=_BF16, device=frame.device)
    dev, st = _ctx(frame)
    t0 = _pb()
    check(L.fosvos_conv7x7s2_first_fwd(frame.data_ptr(), packed.data_ptr(), bias.data_ptr(), y.data_ptr(), n, h, w, co,
                                       (CONV_RELU if relu else 0) | (CONV_FP32_MATH if fp32_math else 0), dev, st),
          "conv7x7s2_first_fwd")
    _pe(t0, "conv7x7s2_first", 2.0 * n * ho * wo * 147 * co, 4 * frame.numel() + 2 * y.numel())
    return y


def conv7x7s2_pool_first_fwd(frame: torch.Tensor, packed: torch.Tensor, bias: torch.Tensor, co: int) -> torch.Tensor:
    """layer_base in one launch (bf16 MFMA form): conv 7x7/2 + folded BatchNorm + ReLU + MaxPool2d(3, 2, 1);
    frame fp32 NCHW [N,3,H,W] -> bf16 NHWC [N,Hp,Wp,ru8(co)]."""
    _need(frame, _F32, "conv7x7s2_pool_first_fwd frame")
    _need(packed, _F32, "conv7x7s2_pool_first_fwd packed weights"); _need(bias, _F32, "conv7x7s2_pool_first_fwd bias")
    n, c, h, w = frame.shape
    L = lib()
    if c != 3:
        raise ValueError("conv7x7s2_pool_first_fwd: 3-channel frames only")
    if packed.numel() != L.fosvos_conv7x7_packed_elems(co) or bias.numel() != L.fosvos_conv2d_bias_elems(co):
        raise ValueError("conv7x7s2_pool_first_fwd: packed image / bias size does not match co")
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    hp, wp = (ho - 1) // 2 + 1, (wo - 1) // 2 + 1
    y = torch.empty((n, hp, wp, _ru(co, 8)), dtype=_BF16, device=frame.device)
    dev, st = _ctx(frame)
    t0 = _pb()
    check(L.fosvos_conv7x7s2_pool_first_fwd(frame.data_ptr(), packed.data_ptr(), bias.data_ptr(), y.data_ptr(), n, h, w, co,
                                            dev, st), "conv7x7s2_pool_first_fwd")
    _pe(t0, "conv7x7s2_pool_first", 2.0 * n * ho * wo * 147 * co, 4 * frame.numel() + 2 * y.numel())
    return y


def maxpool3x3s2_fwd(x: torch.Tensor) -> torch.Tensor:
    _need(x, _BF16, "maxpool3x3s2_fwd")
    n, h, w, c = x.shape
    y = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), dtype=_BF16, device=x.device)
    dev, st = _ctx(x)
    t0 = _pb()
    check(lib().fosvos_maxpool3x3s2_fwd(x.data_ptr(), y.data_ptr(), n, h, w, c, dev, st), "maxpool3x3s2_fwd")
    _pe(t0, "maxpool3x3s2", 0.0, 2 * (x.numel() + y.numel()))
    return y


def deconv_head_fwd(side: Sequence[torch.Tensor], strides: Sequence[int], filt: Sequence[torch.Tensor],
                    filt1: Optional[Sequence[torch.Tensor]], dsn_w: Optional[torch.Tensor],
                    dsn_b: Optional[torch.Tensor], fuse_b: torch.Tensor, H: int, W: int, with_side_out: bool = True):
    """side[s]: fp32 NHWC [N,hs,ws,16]; filt[s]: [2f,2f,16] (upscale filter contracted with the fuse weights);
    filt1[s]: [2f,2f]; dsn_w [4,16]; dsn_b [4]; fuse_b [1].  Returns (fused [N,1,H,W], [4 side outputs] or None)."""
    n = side[0].shape[0]
    for s in range(4):
        _need(side[s], _F32, "deconv_head_fwd side"); _need(filt[s], _F32, "deconv_head_fwd filt")
        k = 2 * int(strides[s])
        if side[s].shape[3] != 16 or tuple(filt[s].shape) != (k, k, 16):
            raise ValueError("deconv_head_fwd: side/filter shape")
    _need(fuse_b, _F32, "deconv_head_fwd fuse_b")
    dev_t = side[0].device
    fused = torch.empty((n, 1, H, W), dtype=_F32, device=dev_t)
    outs = None
    so_ptrs = [None] * 4
    f1_ptrs = [None] * 4
    if with_side_out:
        outs = [torch.empty((n, 1, H, W), dtype=_F32, device=dev_t) for _ in range(4)]
        so_ptrs = [o.data_ptr() for o in outs]
        for s in range(4):
            _need(filt1[s], _F32, "deconv_head_fwd filt1")
            if tuple(filt1[s].shape) != (2 * int(strides[s]),) * 2:
                raise ValueError("deconv_head_fwd: filt1 shape")
        f1_ptrs = [f.data_ptr() for f in filt1]
        _need(dsn_w, _F32, "deconv_head_fwd dsn_w"); _need(dsn_b, _F32, "deconv_head_fwd dsn_b")
    dev, st = _ctx(side[0])
    t0 = _pb()
    check(lib().fosvos_deconv_head_fwd(ptr_array4([t.data_ptr() for t in side]), int_array4([t.shape[1] for t in side]),
                                       int_array4([t.shape[2] for t in side]), int_array4([int(f) for f in strides]),
                                       ptr_array4([t.data_ptr() for t in filt]), ptr_array4(f1_ptrs),
                                       _p(dsn_w) if with_side_out else None, _p(dsn_b) if with_side_out else None,
                                       fuse_b.data_ptr(), fused.data_ptr(), ptr_array4(so_ptrs), n, H, W, dev, st),
          "deconv_head_fwd")
    _pe(t0, "deconv_head_fwd", 2.0 * n * H * W * 256, 4 * (sum(t.numel() for t in side) + n * H * W * (5 if with_side_out else 1)))
    return fused, outs
