"""ctypes binding of libfosvos_hip.so (the C ABI declared in include/fosvos_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, an exception is
raised.  Nothing here imports the CPU oracle.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_size_t, c_uint, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)                       # .../fosvos_amd
# FOSVOS_HIP_LIB: another build of the same ABI (the lab build `make -C fosvos_amd/csrc lab`, whose FOSVOS_* switches the shipped
# library does not read); bench.py records every FOSVOS_* variable of the process in its JSON line
LIB_PATH = os.environ.get("FOSVOS_HIP_LIB") or os.path.join(PKG_ROOT, "lib", "libfosvos_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(PKG_ROOT), "include", "fosvos_hip.h")

ABI_VERSION = 17
CONV_RELU = 1
CONV_OUT_F32 = 2


class FosvosHipError(RuntimeError):
    pass


class SgdEntry(ctypes.Structure):
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("momentum_buf", c_void_p), ("numel", c_int64),
                ("lr", c_float), ("weight_decay", c_float)]


_P4 = c_void_p * 4
_I4 = c_int * 4
_P13 = c_void_p * 13


class VggWeights(ctypes.Structure):
    """fosvos_vgg_weights (include/fosvos_hip.h): device pointers, struct itself in host memory."""
    _fields_ = [("conv_w", _P13), ("conv_b", _P13), ("conv_wf", _P13), ("conv_wd", _P13),
                ("side_b", _P4), ("side_wf", _P4), ("side_wd", _P4), ("filt", _P4), ("filt1", _P4),
                ("dsn_w", c_void_p), ("dsn_b", c_void_p), ("fuse_w", c_void_p), ("fuse_b", c_void_p),
                ("filt_uniform", c_int)]


class VggGrads(ctypes.Structure):
    """fosvos_vgg_grads."""
    _fields_ = [("conv_w", _P13), ("conv_b", _P13), ("side_w", _P4), ("side_b", _P4),
                ("dsn_w", c_void_p), ("dsn_b", c_void_p), ("fuse_w", c_void_p), ("fuse_b", c_void_p),
                ("accumulate", c_int), ("defer_join", c_int), ("bucket_events", c_int), ("last_pass_of_cycle", c_int)]

class Conv3x3PlanInfo(ctypes.Structure):
    """fosvos_conv3x3_plan_info."""
    _fields_ = [("tile_h", c_int), ("tile_w", c_int), ("tile_co", c_int), ("k_splits", c_int), ("workgroups", c_int),
                ("persistent", c_int)]


class ProfileRecord(ctypes.Structure):
    """fosvos_profile_record."""
    _fields_ = [("name", ctypes.c_char * 96), ("launches", c_int), ("ms", ctypes.c_double), ("flops", ctypes.c_double)]


class PackEntry(ctypes.Structure):
    """fosvos_pack_entry."""
    _fields_ = [("w", c_void_p), ("w_fwd", c_void_p), ("w_dgrad", c_void_p), ("Co", c_int), ("Ci", c_int)]



class Conv2dDesc(ctypes.Structure):
    """fosvos_conv2d_desc."""
    _fields_ = [("w_packed", c_void_p), ("bias", c_void_p), ("Ci", c_int), ("Co", c_int), ("k", c_int), ("stride", c_int),
                ("kind", c_int)]


class ResnetBlock(ctypes.Structure):
    """fosvos_resnet_block."""
    _fields_ = [("conv", Conv2dDesc * 3), ("n_convs", c_int), ("has_down", c_int), ("down", Conv2dDesc)]


class ResnetNet(ctypes.Structure):
    """fosvos_resnet_net (host struct; `blocks` points at a host array of ResnetBlock)."""
    _fields_ = [("first_w", c_void_p), ("first_b", c_void_p), ("first_co", c_int), ("first_fp32_math", c_int), ("first_unfused", c_int),
                ("blocks_per_stage", c_int * 4),
                ("blocks", POINTER(ResnetBlock)), ("side", Conv2dDesc * 4), ("filt", c_void_p * 4),
                ("filt1", c_void_p * 4), ("stride", c_int * 4), ("dsn_w", c_void_p), ("dsn_b", c_void_p),
                ("fuse_b", c_void_p)]

# name -> (restype, argtypes); every entry point of include/fosvos_hip.h
SIGNATURES = {
    "fosvos_abi_version": (c_int, []),
    "fosvos_last_error": (c_char_p, []),
    "fosvos_build_arch": (c_char_p, []),
    "fosvos_ctx_create": (c_int, [c_int, POINTER(c_void_p)]),
    "fosvos_ctx_destroy": (c_int, [c_void_p]),
    "fosvos_ctx_device": (c_int, [c_void_p]),
    "fosvos_conv3x3_plan": (c_int, [c_int, c_int, c_int, c_int, c_int, POINTER(Conv3x3PlanInfo)]),
    "fosvos_conv3x3_fwd_plan": (c_int, [c_int, c_int, c_int, c_int, c_int, ctypes.c_uint, POINTER(Conv3x3PlanInfo)]),
    "fosvos_conv3x3_first_plan": (c_int, [c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "fosvos_nchw_f32_to_nhwc_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fosvos_nhwc_bf16_to_nchw_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fosvos_nhwc_f32_to_nchw_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fosvos_nchw_f32_to_nhwc_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fosvos_pack_conv3x3_weights": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "fosvos_packed_weight_elems": (c_size_t, [c_int, c_int]),
    "fosvos_pack_conv3x3_weights_multi": (c_int, [POINTER(PackEntry), c_int, c_int, c_void_p]),
    "fosvos_conv3x3_first_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                         c_void_p]),
    "fosvos_conv3x3_first_fwd_bits": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                              c_void_p]),
    "fosvos_conv3x3_dgrad_bits": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                          c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "fosvos_conv3x3_dgrad_unpool": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                            c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "fosvos_conv3x3_first_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                           c_void_p, c_size_t, c_int, c_void_p]),
    "fosvos_conv3x3_first_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "fosvos_conv3x3_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_uint,
                                   c_void_p, c_size_t, c_int, c_void_p]),
    "fosvos_conv3x3_fwd_pool": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                        c_int, c_uint, c_void_p, c_size_t, c_int, c_void_p]),
    "fosvos_conv3x3_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "fosvos_conv3x3_dgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                     c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "fosvos_conv3x3_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                     c_void_p, c_size_t, c_int, c_void_p]),
    "fosvos_conv3x3_wgrad_slabs": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
                                           c_int, c_void_p]),
    "fosvos_conv3x3_wgrad_reduce": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                            c_size_t, c_int, c_void_p]),
    "fosvos_conv3x3_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "fosvos_maxpool2x2_ceil_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fosvos_maxpool2x2_ceil_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                           c_void_p]),
    "fosvos_head_fwd": (c_int, [POINTER(c_void_p), POINTER(c_int), POINTER(c_int), POINTER(c_void_p),
                                POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fosvos_head_bwd": (c_int, [POINTER(c_void_p), POINTER(c_int), POINTER(c_int), POINTER(c_void_p),
                                POINTER(c_void_p), c_void_p, c_void_p, c_void_p, POINTER(c_void_p),
                                POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                c_void_p, c_size_t, c_int, c_void_p]),
    "fosvos_head_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "fosvos_cbce_loss": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p, c_size_t,
                                 c_int, c_void_p]),
    "fosvos_cbce_workspace_bytes": (c_size_t, [c_int64]),
    "fosvos_cbce_loss_frames": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                        c_size_t, c_int, c_void_p]),
    "fosvos_cbce_loss_frames_parts": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p, c_void_p,
                                              c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "fosvos_cbce_loss_batch_counts": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_size_t, c_int, c_void_p]),
    "fosvos_sgd_momentum_step": (c_int, [c_void_p, c_int, c_int64, c_float, c_int, c_int, c_void_p]),
    "fosvos_conv3x3_s2_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_uint,
                                      c_void_p, c_size_t, c_int, c_void_p]),
    "fosvos_conv3x3_fwd_add": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                       c_uint, c_void_p, c_size_t, c_int, c_void_p]),
    "fosvos_fold_conv_bn": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_float, c_void_p, c_void_p, c_int, c_void_p]),
    "fosvos_conv2d_packed_dwords": (c_size_t, [c_int, c_int, c_int]),
    "fosvos_conv2d_bias_elems": (c_size_t, [c_int]),
    "fosvos_pack_conv2d_bn": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_float, c_void_p, c_void_p, c_int, c_void_p]),
    "fosvos_conv2d_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_uint, c_int, c_void_p]),
    "fosvos_conv7x7_packed_elems": (c_size_t, [c_int]),
    "fosvos_pack_conv7x7_bn": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                       c_void_p, c_int, c_void_p]),
    "fosvos_conv7x7s2_first_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_uint,
                                           c_int, c_void_p]),
    "fosvos_conv7x7s2_pool_first_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                                c_void_p]),
    "fosvos_maxpool3x3s2_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fosvos_deconv_head_fwd": (c_int, [POINTER(c_void_p), POINTER(c_int), POINTER(c_int), POINTER(c_int),
                                       POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p,
                                       POINTER(c_void_p), c_int, c_int, c_int, c_int, c_void_p]),
    "fosvos_resnet_arena_bytes": (c_size_t, [POINTER(ResnetNet), c_int, c_int, c_int]),
    "fosvos_resnet_forward": (c_int, [POINTER(ResnetNet), c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p,
                                      POINTER(c_void_p), c_int, c_void_p, c_void_p, c_void_p]),
    "fosvos_vgg_arena_bytes": (c_size_t, [c_int, c_int, c_int]),
    "fosvos_vgg_grad_bucket_wait": (c_int, [c_void_p, c_int, c_void_p]),
    "fosvos_profile_start": (c_int, [c_int, c_int]),
    "fosvos_profile_stop": (c_int, [c_int, POINTER(ProfileRecord), c_int, POINTER(c_int)]),
    "fosvos_vgg_forward": (c_int, [POINTER(VggWeights), c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p,
                                   POINTER(c_void_p), c_int, c_void_p]),
    "fosvos_vgg_forward_streams": (c_int, [c_void_p, POINTER(VggWeights), c_void_p, c_int, c_int, c_int, c_void_p, c_size_t,
                                           c_void_p, POINTER(c_void_p), c_void_p, c_void_p]),
    "fosvos_vgg_backward": (c_int, [c_void_p, POINTER(VggWeights), POINTER(VggGrads), c_void_p, c_int, c_int, c_int, c_void_p,
                                    c_size_t, c_void_p, POINTER(c_void_p), c_void_p, c_void_p]),
}

_lib = None
_lock = threading.Lock()


def lib() -> ctypes.CDLL:
    """Load (once) and return the shared library; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise FosvosHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C fosvos_amd/csrc`.  There is no CPU fallback.")
        # torch first: its HIP runtime must be the one in the process when the library's own dependency on libamdhip64 is
        # resolved (loaded the other way round - the library before torch, as a build()-then-smoke() process did - the
        # library's calls ran on a second runtime that found no device: "no ROCm-capable device is detected")
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing
            fn.restype = restype
            fn.argtypes = argtypes
        got = handle.fosvos_abi_version()
        if got != ABI_VERSION:
            raise FosvosHipError(f"libfosvos_hip.so ABI {got} != binding ABI {ABI_VERSION}: rebuild")
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().fosvos_last_error()
        raise FosvosHipError(f"{what} failed (code {rc}): {msg.decode(errors='replace') if msg else ''}")


class LaunchProfile:
    """``with LaunchProfile(device_index) as prof: ...`` brackets every kernel the library launches inside the block
    with timing events on its own launch stream; afterwards ``prof.records`` maps kernel name -> dict(launches, ms,
    flops)."""

    def __init__(self, device: int, max_launches: int = 65536):
        self.device, self.max_launches, self.records = int(device), int(max_launches), {}

    def __enter__(self):
        check(lib().fosvos_profile_start(self.device, self.max_launches), "profile_start")
        return self

    def __exit__(self, *exc):
        buf = (ProfileRecord * 128)()
        n = c_int(0)
        check(lib().fosvos_profile_stop(self.device, buf, 128, ctypes.byref(n)), "profile_stop")
        self.records = {buf[i].name.decode(): {"launches": buf[i].launches, "ms": buf[i].ms, "flops": buf[i].flops}
                        for i in range(n.value)}
        return False


class Context:
    """A caller-owned fosvos_ctx (include/fosvos_hip.h): the inter-stream events of one model's multi-stream passes and
    its gradient-bucket state.  One per model and device; destroyed with its owner."""

    def __init__(self, device: int):
        h = c_void_p()
        check(lib().fosvos_ctx_create(int(device), ctypes.byref(h)), "ctx_create")
        self.handle, self.device = h, int(device)

    def close(self) -> None:
        h, self.handle = self.handle, None
        if h is not None and _lib is not None:
            _lib.fosvos_ctx_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: the process is going away with its events
            pass


def ptr_array4(ptrs) -> "ctypes.Array":
    return _P4(*[c_void_p(p) for p in ptrs])


def int_array4(vals) -> "ctypes.Array":
    return _I4(*[int(v) for v in vals])
