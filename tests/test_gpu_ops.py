"""Per-kernel parity on a real MI355X: every C-ABI entry point against the CPU oracle (plain torch
fp32 on the same, identically rounded inputs).  Run with ``pytest -m gpu``.

Tolerances (stated per test):
  * bf16 outputs: one bf16 rounding of an fp32-accumulated value: |err| <= 2^-8 |ref| + fp32 noise
  * fp32 outputs: fp32 accumulation-order noise only
  * integer/index work (pool selection and routing): exact
"""
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import osvos_ref as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from fosvos_hip import ops as _ops
    return _ops


def bf(t):  # round to bf16, keep fp32 container
    return t.to(torch.bfloat16).float()


def to_nhwc_bf16(t):  # fp32 NCHW (cpu) -> bf16 NHWC (gpu), via torch (test plumbing)
    return t.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def from_nhwc(t):  # bf16/fp32 NHWC (gpu) -> fp32 NCHW (cpu)
    return t.float().permute(0, 3, 1, 2).contiguous().cpu()


def rel_err(a, ref):
    return (a - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)


def assert_bf16_close(a, ref, name, extra=0.0):
    """a is a bf16-rounded version of something that should equal ref up to fp32 accumulation noise."""
    tol = (2.0 ** -8) * ref.abs() + (1e-5 + extra) * ref.abs().max()
    bad = (a - ref).abs() > tol
    assert not bad.any(), f"{name}: {int(bad.sum())} / {bad.numel()} outside bf16 tolerance, max rel-to-max err {rel_err(a, ref):.3e}"


def gen(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ------------------------------------------------------------------------------------------ layout
def test_layout_roundtrip(ops):
    x = gen(2, 5, 7, 9, seed=1)
    y = ops.nchw_to_nhwc_bf16(x.to(DEV), c_pad=8)
    assert y.shape == (2, 7, 9, 8)
    assert torch.equal(from_nhwc(y)[:, :5], bf(x))
    assert torch.equal(from_nhwc(y)[:, 5:], torch.zeros(2, 3, 7, 9))
    back = ops.nhwc_bf16_to_nchw(y, c=5).cpu()
    assert torch.equal(back, bf(x))
    s = gen(2, 16, 5, 6, seed=2).to(DEV)
    assert torch.equal(ops.nhwc_f32_to_nchw(ops.nchw_to_nhwc_f32(s)), s)
    assert torch.equal(ops.nchw_to_nhwc_f32(s).cpu(), s.cpu().permute(0, 2, 3, 1).contiguous())


def _pack_ref(w, transpose):
    """Python statement of the packed image [K32][9][4][out_pad][8] (include/fosvos_hip.h)."""
    co, ci = w.shape[:2]
    wf = w.reshape(co, ci, 9)
    if transpose:  # (out=ci, in=co, tap' = 8 - tap)
        wf = wf.flip(2).permute(1, 0, 2)
    out_ch, in_ch = wf.shape[:2]
    out_pad, in_pad = (out_ch + 15) // 16 * 16, (in_ch + 31) // 32 * 32
    p = torch.zeros(out_pad, in_pad, 9)
    p[:out_ch, :in_ch] = wf
    p = p.reshape(out_pad, in_pad // 32, 4, 8, 9).permute(1, 4, 2, 0, 3).contiguous()  # [K32][9][4][out][8]
    return p.reshape(-1)


@pytest.mark.parametrize("co,ci", [(64, 64), (16, 128), (128, 64), (64, 3), (16, 40)])
def test_pack_weights(ops, co, ci):
    w = gen(co, ci, 3, 3, seed=3)
    fwd, dgr = ops.pack_conv3x3_weights(w.to(DEV))
    assert torch.equal(fwd.float().cpu(), bf(_pack_ref(w, False)))
    assert torch.equal(dgr.float().cpu(), bf(_pack_ref(w, True)))


@pytest.mark.parametrize("shapes,offset", [
    ([(64, 64), (128, 64), (16, 128), (256, 128), (16, 512), (96, 32)], 0),
    ([(64, 64), (128, 64), (128, 128), (16, 128)], 0),   # what the late half of the split optimizer step repacks: 8-channel blocks
    ([(512, 512), (16, 512), (256, 128), (40, 64)], 0),  # > 256 blocks: 32-channel blocks
    ([(64, 64), (16, 128)], 1), ([(512, 256), (16, 256)], 3),  # masters that are not 16-byte aligned (views into a flat buffer)
])
def test_pack_weights_multi(ops, shapes, offset):
    """One launch for many layers (what the module does after an optimizer step) == the per-layer packer, bit for bit;
    covers a 16-channel side layer (half-empty 32-channel block), non-square layers, both block sizes of the kernel and
    masters at any 4-byte alignment."""
    ws = [gen(co, ci, 3, 3, seed=40 + i) for i, (co, ci) in enumerate(shapes)]

    def on_device(w):
        if not offset:
            return w.to(DEV)
        flat = torch.empty(w.numel() + offset, device=DEV)
        flat[offset:] = w.reshape(-1).to(DEV)
        v = flat[offset:].view(w.shape)
        assert v.data_ptr() % 16 == 4 * offset and v.is_contiguous()
        return v
    got = ops.pack_conv3x3_weights_multi([on_device(w) for w in ws])
    assert len(got) == len(ws)
    for w, (fwd, dgr) in zip(ws, got):
        assert torch.equal(fwd.float().cpu(), bf(_pack_ref(w, False)))
        assert torch.equal(dgr.float().cpu(), bf(_pack_ref(w, True)))


# ------------------------------------------------------------------------------------------ conv1_1
@pytest.mark.parametrize("n,h,w", [(1, 48, 86), (2, 61, 107), (1, 5, 3), (1, 4, 64), (1, 9, 130)])
def test_conv_first_fwd(ops, n, h, w):
    x = gen(n, 3, h, w, seed=4, scale=60.0)
    wt = gen(64, 3, 3, 3, seed=5, scale=0.2)
    b = gen(64, seed=6, scale=0.5)
    ref = F.relu(F.conv2d(x, wt, b, padding=1))
    y = ops.conv3x3_first_fwd(x.to(DEV), wt.to(DEV), b.to(DEV))
    assert y.shape == (n, h, w, 64) and y.dtype == torch.bfloat16
    assert_bf16_close(from_nhwc(y), ref, "conv_first_fwd")


@pytest.mark.parametrize("n,h,w", [(1, 48, 86), (2, 33, 47), (1, 3, 300)])
def test_conv_first_wgrad(ops, n, h, w):
    x = gen(n, 3, h, w, seed=7, scale=60.0)
    dy = bf(gen(n, 64, h, w, seed=8))
    wt = torch.zeros(64, 3, 3, 3, requires_grad=True)
    b = torch.zeros(64, requires_grad=True)
    # the kernel contracts a bf16 image of the frame (like every other wgrad reads bf16 activations), fp32 accumulate
    F.conv2d(bf(x), wt, b, padding=1).backward(dy)
    dw, db = ops.conv3x3_first_wgrad(x.to(DEV), to_nhwc_bf16(dy))
    assert rel_err(dw.cpu(), wt.grad) < 5e-5
    assert rel_err(db.cpu(), b.grad) < 5e-5
    # and it stays within bf16 input rounding of the fp32-frame gradient
    wt2 = torch.zeros(64, 3, 3, 3, requires_grad=True)
    F.conv2d(x, wt2, None, padding=1).backward(dy)
    assert rel_err(dw.cpu(), wt2.grad) < 5e-3


# ------------------------------------------------------------------------------------------ MFMA conv
CONV_CASES = [
    # n, h, w, ci, co  -> exercises each tile configuration and ragged edges
    (1, 64, 96, 64, 64),     # TileBig
    (1, 61, 107, 64, 128),   # TileBig/Mid, ragged
    (2, 30, 54, 128, 64),    # TileMid/Small
    (1, 8, 16, 64, 64),      # single small tile
    (1, 15, 27, 256, 128),   # TileSmall, 8 K chunks
    (1, 7, 5, 64, 64),       # smaller than one tile
    (1, 1, 1, 64, 64),       # degenerate
    (1, 120, 214, 64, 64),   # 105 workgroups of 256 px: still the 128-pixel tile (the 256-px tiles: HOT_CASES below)
]


@pytest.mark.parametrize("n,h,w,ci,co", CONV_CASES)
def test_conv3x3_fwd(ops, n, h, w, ci, co):
    x = bf(gen(n, ci, h, w, seed=10))
    wt = gen(co, ci, 3, 3, seed=11, scale=math.sqrt(2.0 / (9 * ci)))
    b = gen(co, seed=12, scale=0.2)
    ref = F.relu(F.conv2d(x, bf(wt), b, padding=1))
    wf, _ = ops.pack_conv3x3_weights(wt.to(DEV))
    y = ops.conv3x3_fwd(to_nhwc_bf16(x), wf, b.to(DEV), ci, co, relu=True)
    assert y.shape == (n, h, w, co)
    assert_bf16_close(from_nhwc(y), ref, f"conv3x3_fwd {n}x{h}x{w} {ci}->{co}")
    # no-ReLU, no-bias variant
    ref2 = F.conv2d(x, bf(wt), None, padding=1)
    y2 = ops.conv3x3_fwd(to_nhwc_bf16(x), wf, None, ci, co, relu=False)
    assert_bf16_close(from_nhwc(y2), ref2, "conv3x3_fwd linear")


@pytest.mark.parametrize("n,h,w,ci,co", [(1, 48, 86, 64, 64), (1, 61, 107, 64, 128), (2, 33, 47, 128, 128),
                                          (1, 15, 27, 256, 512), (1, 7, 5, 64, 64), (1, 120, 214, 64, 64)])
def test_conv3x3_fwd_pool(ops, n, h, w, ci, co):
    """Last conv of a stage with the ceil-mode pool fused into its epilogue: y identical to the plain conv, pooled map
    identical to the pool kernel on y (odd sizes: ragged last row/column windows; small maps: the split-K fallback)."""
    x = bf(gen(n, ci, h, w, seed=50))
    wt = gen(co, ci, 3, 3, seed=51, scale=math.sqrt(2.0 / (9 * ci)))
    b = gen(co, seed=52, scale=0.2)
    wf, _ = ops.pack_conv3x3_weights(wt.to(DEV))
    y_ref = ops.conv3x3_fwd(to_nhwc_bf16(x), wf, b.to(DEV), ci, co, relu=True)
    y, yp = ops.conv3x3_fwd_pool(to_nhwc_bf16(x), wf, b.to(DEV), ci, co, relu=True)
    assert torch.equal(y, y_ref)
    assert yp.shape == (n, (h + 1) // 2, (w + 1) // 2, co)
    assert torch.equal(yp, ops.maxpool_fwd(y_ref))
    assert torch.equal(from_nhwc(yp), F.max_pool2d(from_nhwc(y_ref), 2, 2, ceil_mode=True))


@pytest.mark.parametrize("n,h,w,ci", [(1, 120, 214, 128), (1, 30, 54, 512), (2, 15, 27, 256), (1, 3, 2, 128),
                                       (1, 300, 300, 128)])
def test_conv3x3_side_prep_f32(ops, n, h, w, ci):
    """16-channel fp32-output variant (side_prep, no ReLU)."""
    x = bf(gen(n, ci, h, w, seed=13))
    wt = gen(16, ci, 3, 3, seed=14, scale=math.sqrt(1.0 / (9 * ci)))
    b = gen(16, seed=15, scale=0.2)
    ref = F.conv2d(x, bf(wt), b, padding=1)
    wf, _ = ops.pack_conv3x3_weights(wt.to(DEV))
    y = ops.conv3x3_fwd(to_nhwc_bf16(x), wf, b.to(DEV), ci, 16, relu=False, out_f32=True)
    assert y.dtype == torch.float32 and y.shape == (n, h, w, 16)
    assert rel_err(from_nhwc(y), ref) < 2e-5


def test_conv3x3_fwd_scaling_full_size(ops):
    """Size-independent property at a BASELINE-size map (240x427, stage 2): conv(2a) == 2 conv(a) bit
    for bit (a power-of-two scale commutes with every rounding), and conv(0) == 0."""
    ci, co, h, w = 64, 64, 240, 427
    a = bf(gen(1, ci, h, w, seed=16))
    wt = gen(co, ci, 3, 3, seed=18, scale=0.05)
    wf, _ = ops.pack_conv3x3_weights(wt.to(DEV))
    ya = ops.conv3x3_fwd(to_nhwc_bf16(a), wf, None, ci, co, relu=False).float()
    y2 = ops.conv3x3_fwd(to_nhwc_bf16(2 * a), wf, None, ci, co, relu=False).float()
    assert torch.equal(y2, 2 * ya)
    y0 = ops.conv3x3_fwd(to_nhwc_bf16(torch.zeros_like(a)), wf, None, ci, co, relu=False).float()
    assert not y0.any()
    # spot-check 2000 random outputs against the oracle arithmetic
    g = torch.Generator().manual_seed(19)
    ys = torch.randint(0, h, (2000,), generator=g)
    xs = torch.randint(0, w, (2000,), generator=g)
    cs = torch.randint(0, co, (2000,), generator=g)
    ap = F.pad(a, [1, 1, 1, 1])
    wb = bf(wt)
    ref = torch.stack([(ap[0, :, y:y + 3, x:x + 3] * wb[c]).sum() for y, x, c in zip(ys.tolist(), xs.tolist(), cs.tolist())])
    got = ya.cpu()[0, ys, xs, cs]
    assert_bf16_close(got, ref, "full-size spot check")


@pytest.mark.parametrize("n,h,w,ci,co", [(1, 64, 96, 64, 64), (1, 61, 107, 64, 128), (2, 30, 54, 128, 64),
                                          (1, 15, 27, 256, 128), (1, 7, 5, 64, 64)])
def test_conv3x3_dgrad(ops, n, h, w, ci, co):
    dy = bf(gen(n, co, h, w, seed=20))
    wt = gen(co, ci, 3, 3, seed=21, scale=math.sqrt(2.0 / (9 * ci)))
    xin = torch.zeros(n, ci, h, w, requires_grad=True)
    F.conv2d(xin, bf(wt), None, padding=1).backward(dy)
    ref = xin.grad
    _, wd = ops.pack_conv3x3_weights(wt.to(DEV))
    dx = ops.conv3x3_dgrad(to_nhwc_bf16(dy), wd, ci, co)
    assert_bf16_close(from_nhwc(dx), ref, "conv3x3_dgrad")
    # fused ReLU mask + addend (addend aliasing the output)
    act = bf(F.relu(gen(n, ci, h, w, seed=22)))
    add = bf(gen(n, ci, h, w, seed=23))
    ref2 = bf(bf(ref) * (act > 0).float() + add)
    add_dev = to_nhwc_bf16(add)
    dx2 = ops.conv3x3_dgrad(to_nhwc_bf16(dy), wd, ci, co, relu_src=to_nhwc_bf16(act), addend=add_dev, out=add_dev)
    assert dx2.data_ptr() == add_dev.data_ptr()
    got = from_nhwc(dx2)
    # two roundings: the intermediate may land on the neighbouring bf16 (1 ulp <= 2^-7 relative)
    tol = (2.0 ** -7) * ref2.abs() + 1e-5 * ref2.abs().max() + (2.0 ** -7) * ref.abs()
    assert ((got - ref2).abs() <= tol).all(), f"dgrad mask+add: rel err {rel_err(got, ref2):.3e}"


def test_conv3x3_dgrad_side(ops):
    """side_prep dgrad: 16 real channels zero-padded to 32 on the contraction side."""
    n, h, w, ci, co = 1, 30, 54, 128, 16
    dy = bf(gen(n, co, h, w, seed=24))
    wt = gen(co, ci, 3, 3, seed=25, scale=0.05)
    xin = torch.zeros(n, ci, h, w, requires_grad=True)
    F.conv2d(xin, bf(wt), None, padding=1).backward(dy)
    _, wd = ops.pack_conv3x3_weights(wt.to(DEV))
    dy_pad = torch.zeros(n, h, w, 32, dtype=torch.bfloat16, device=DEV)
    dy_pad[..., :16] = to_nhwc_bf16(dy)
    dx = ops.conv3x3_dgrad(dy_pad, wd, ci, co)
    assert_bf16_close(from_nhwc(dx), xin.grad, "conv3x3_dgrad side")


@pytest.mark.parametrize("n,h,w,ci,co", [(1, 64, 96, 64, 64), (1, 61, 107, 64, 128), (2, 30, 54, 128, 64),
                                          (1, 15, 27, 256, 128), (1, 7, 5, 64, 64), (1, 17, 33, 128, 16),
                                          (1, 120, 214, 64, 64), (1, 30, 54, 512, 512)])
def test_conv3x3_wgrad(ops, n, h, w, ci, co):
    x = bf(gen(n, ci, h, w, seed=30))
    dy = bf(gen(n, co, h, w, seed=31))
    wt = torch.zeros(co, ci, 3, 3, requires_grad=True)
    b = torch.zeros(co, requires_grad=True)
    F.conv2d(x, wt, b, padding=1).backward(dy)
    cy = (co + 31) // 32 * 32
    dy_dev = torch.zeros(n, h, w, cy, dtype=torch.bfloat16, device=DEV)
    dy_dev[..., :co] = to_nhwc_bf16(dy)
    dw, db = ops.conv3x3_wgrad(to_nhwc_bf16(x), dy_dev, ci, co)
    assert dw.shape == (co, ci, 3, 3)
    assert rel_err(dw.cpu(), wt.grad) < 5e-5, f"wgrad rel err {rel_err(dw.cpu(), wt.grad):.3e}"
    assert rel_err(db.cpu(), b.grad) < 5e-5
    # accumulate mode and run-to-run determinism (fixed-order slab reduction)
    dw2, db2 = ops.conv3x3_wgrad(to_nhwc_bf16(x), dy_dev, ci, co, dw=dw.clone(), db=db.clone(), accumulate=True)
    assert torch.equal(dw2, 2 * dw) and torch.equal(db2, 2 * db)
    dw3, _ = ops.conv3x3_wgrad(to_nhwc_bf16(x), dy_dev, ci, co)
    # the one-call entry point (kernel + reduction) == the split calls the wrapper above uses, bit for bit
    dw4, db4 = ops.conv3x3_wgrad_one_call(to_nhwc_bf16(x), dy_dev, ci, co)
    assert torch.equal(dw4, dw) and torch.equal(db4, db)
    assert torch.equal(dw3, dw)


# ------------------------------------------------------------------------------------------ the instantiations the 480x854 step runs
# The 256-pixel tiles (8x32 and 16x16 x 64 channels) are only selected when a launch has >= 256 of them: these are the kernels
# behind 47 % of the fine-tune step's time (five frames per launch).  Each case states the tile fosvos_conv3x3_plan must
# report, so a later change of the selection rule cannot silently move these tests onto other kernels.
HOT_CASES = [
    # n, h, w, ci, co, tile
    (1, 240, 427, 64, 128, (16, 16, 64)),   # conv2_1 of one frame: 16x16 tiles, ragged right edge (427 = 26 * 16 + 11)
    (5, 120, 214, 128, 128, (8, 32, 64)),   # five frames per launch, stage-3 map: 8x32 tiles, 214 = 6 * 32 + 22
    (1, 480, 854, 64, 64, (8, 32, 64)),     # conv1_2 at full size: 1620 workgroups, XCD-swizzled 1-D grid
    (5, 60, 107, 256, 256, (16, 16, 64)),   # five frames, stage-4 map: 16x16 tiles overhang 60x107 less than 8x32 ones
    (5, 30, 54, 512, 512, (8, 32, 64)),     # five frames, stage 5 as the backward pass runs it: 320 workgroups of 256 pixels
    (3, 30, 54, 512, 512, (8, 16, 64)),     # ... and as the forward pass's three-frame chain does: 384 128-pixel tiles
]


def _assert_tile(ops, n, h, w, in_ch, out_ch, tile):
    plan = ops.conv3x3_plan(n, h, w, in_ch, out_ch)
    assert plan["tile"] == tile and plan["k_splits"] == 1, f"{(n, h, w, in_ch, out_ch)} runs {plan}, the test is about {tile}"
    assert plan["workgroups"] >= 256


# ------------------------------------------------------------------------------------------ the persistent forward kernel
# k_conv3x3_pp (csrc/conv_pp.hip): 256 persistent workgroups of two four-wave groups.  The cases cover: both K-chunk regimes
# (<= 64 input channels: the weight chunks stay in LDS; more: the double-buffered reload), 1 / 2 / 4 / 8 channel blocks per
# XCD, tile counts that leave group 1 of some workgroups without a last tile, ragged right / bottom edges, odd H and W
# (ragged pooling windows), several images per launch.  Each asserts through fosvos_conv3x3_fwd_plan that the launch IS the
# persistent kernel.
PP_CASES = [
    # n, h, w, ci, co
    (1, 480, 854, 64, 64),     # conv1_2 of one frame: 1620 tiles (3.2 rounds: the last round leaves groups without a tile), 854 = 26 * 32 + 22
    (2, 240, 427, 64, 128),    # conv2_1: two channel blocks, 427 = 13 * 32 + 11
    (3, 240, 427, 128, 128),   # four K chunks: the weight chunks are reloaded every super-step
    (4, 60, 107, 256, 256),    # eight K chunks, four channel blocks, exactly one tile per group, bottom tiles half outside (60 = 7 * 8 + 4)
    (2, 60, 107, 256, 512),    # eight channel blocks
    (6, 121, 213, 64, 64),     # odd H and W: the last pooling row / column windows hold one pixel row / column
    (7, 97, 335, 96, 64),      # three K chunks (odd count: chunk index and buffer parity drift apart)
]


@pytest.mark.parametrize("n,h,w,ci,co", PP_CASES)
def test_conv3x3_fwd_persistent(ops, n, h, w, ci, co):
    """Forward conv + bias + ReLU on the persistent kernel, whole tensors against torch fp32 on identically rounded inputs;
    the same launch with the fused ceil-mode pool: y bit-identical, pooled map exactly the pool of y; the linear form (no
    bias, no ReLU); a second launch bit-identical (fixed summation order)."""
    plan = ops.conv3x3_fwd_plan(n, h, w, ci, co)
    assert plan["persistent"] and plan["tile"] == (8, 32, 64) and plan["workgroups"] == 256, plan
    x = bf(gen(n, ci, h, w, seed=150))
    wt = gen(co, ci, 3, 3, seed=151, scale=math.sqrt(2.0 / (9 * ci)))
    b = gen(co, seed=152, scale=0.2)
    ref = F.relu(F.conv2d(x, bf(wt), b, padding=1))
    wf, _ = ops.pack_conv3x3_weights(wt.to(DEV))
    xd = to_nhwc_bf16(x)
    y = ops.conv3x3_fwd(xd, wf, b.to(DEV), ci, co, relu=True)
    assert_bf16_close(from_nhwc(y), ref, f"conv3x3 persistent fwd {n}x{h}x{w} {ci}->{co}")
    y2, yp = ops.conv3x3_fwd_pool(xd, wf, b.to(DEV), ci, co, relu=True)
    assert torch.equal(y2, y)
    assert yp.shape == (n, (h + 1) // 2, (w + 1) // 2, co)
    assert torch.equal(from_nhwc(yp), F.max_pool2d(from_nhwc(y), 2, 2, ceil_mode=True))
    assert ops.conv3x3_fwd_plan(n, h, w, ci, co, relu=False)["persistent"]
    y3 = ops.conv3x3_fwd(xd, wf, None, ci, co, relu=False)
    assert_bf16_close(from_nhwc(y3), F.conv2d(x, bf(wt), None, padding=1), "conv3x3 persistent fwd linear")
    assert torch.equal(ops.conv3x3_fwd(xd, wf, b.to(DEV), ci, co, relu=True), y)


@pytest.mark.parametrize("n,h,w,ci,co,tile", HOT_CASES)
def test_conv3x3_fwd_hot_tiles(ops, n, h, w, ci, co, tile):
    """Forward conv + bias + ReLU on the 256-pixel tiles, whole tensors against torch fp32 on identically rounded inputs;
    the same launch with the fused ceil-mode pool: y bit-identical, pooled map exactly the pool of y.  (Where the forward
    entry point now takes the persistent kernel for this shape, this test covers that; the igemm tiles stay covered by the
    data-gradient tests below, which run the same instantiations.)"""
    fplan = ops.conv3x3_fwd_plan(n, h, w, ci, co)
    if not fplan["persistent"]:
        _assert_tile(ops, n, h, w, ci, co, tile)
    x = bf(gen(n, ci, h, w, seed=110))
    wt = gen(co, ci, 3, 3, seed=111, scale=math.sqrt(2.0 / (9 * ci)))
    b = gen(co, seed=112, scale=0.2)
    ref = F.relu(F.conv2d(x, bf(wt), b, padding=1))
    wf, _ = ops.pack_conv3x3_weights(wt.to(DEV))
    xd = to_nhwc_bf16(x)
    y = ops.conv3x3_fwd(xd, wf, b.to(DEV), ci, co, relu=True)
    assert_bf16_close(from_nhwc(y), ref, f"conv3x3_fwd {tile} {n}x{h}x{w} {ci}->{co}")
    y2, yp = ops.conv3x3_fwd_pool(xd, wf, b.to(DEV), ci, co, relu=True)
    assert torch.equal(y2, y)
    assert torch.equal(from_nhwc(yp), F.max_pool2d(from_nhwc(y), 2, 2, ceil_mode=True))
    # no ReLU: negative values through the pool's float path
    y3, yp3 = ops.conv3x3_fwd_pool(xd, wf, b.to(DEV), ci, co, relu=False)
    assert_bf16_close(from_nhwc(y3), F.conv2d(x, bf(wt), b, padding=1), "conv3x3_fwd_pool linear")
    assert torch.equal(from_nhwc(yp3), F.max_pool2d(from_nhwc(y3), 2, 2, ceil_mode=True))


@pytest.mark.parametrize("n,h,w,ci,co,tile", HOT_CASES)
def test_conv3x3_dgrad_hot_tiles(ops, n, h, w, ci, co, tile):
    """Data gradient on the 256-pixel tiles (contraction over the forward op's outputs, so the plan is queried with the
    channel roles swapped): plain, and with the ReLU mask of the producer + the other consumer's gradient added in place
    (the epilogue that reads both through buffer descriptors)."""
    # the tile of the dgrad launch: in = co, out = ci
    plan = ops.conv3x3_plan(n, h, w, co, ci)
    assert plan["tile"][:2] == tile[:2] and plan["tile"][2] == 64 and plan["k_splits"] == 1 and plan["workgroups"] >= 256, plan
    dy = bf(gen(n, co, h, w, seed=120))
    wt = gen(co, ci, 3, 3, seed=121, scale=math.sqrt(2.0 / (9 * ci)))
    xin = torch.zeros(n, ci, h, w, requires_grad=True)
    F.conv2d(xin, bf(wt), None, padding=1).backward(dy)
    ref = xin.grad
    _, wd = ops.pack_conv3x3_weights(wt.to(DEV))
    dyd = to_nhwc_bf16(dy)
    dx = ops.conv3x3_dgrad(dyd, wd, ci, co)
    assert_bf16_close(from_nhwc(dx), ref, f"conv3x3_dgrad {tile}")
    act = bf(F.relu(gen(n, ci, h, w, seed=122)))
    add = bf(gen(n, ci, h, w, seed=123))
    ref2 = bf(bf(ref) * (act > 0).float() + add)
    add_dev = to_nhwc_bf16(add)
    dx2 = ops.conv3x3_dgrad(dyd, wd, ci, co, relu_src=to_nhwc_bf16(act), addend=add_dev, out=add_dev)
    assert dx2.data_ptr() == add_dev.data_ptr()
    got = from_nhwc(dx2)
    tol = (2.0 ** -7) * ref2.abs() + 1e-5 * ref2.abs().max() + (2.0 ** -7) * ref.abs()
    assert ((got - ref2).abs() <= tol).all(), f"dgrad mask+add {tile}: rel err {rel_err(got, ref2):.3e}"
    # mask only (what the dgrad of a conv inside a stage runs)
    dx3 = ops.conv3x3_dgrad(dyd, wd, ci, co, relu_src=to_nhwc_bf16(act))
    assert torch.equal(from_nhwc(dx3), from_nhwc(dx) * (act > 0).float())


@pytest.mark.parametrize("n,h,w,ci,tile", [(1, 240, 427, 128, (16, 16, 64)), (5, 120, 214, 256, (8, 32, 64))])
def test_conv3x3_dgrad_side_hot_tiles(ops, n, h, w, ci, tile):
    """side_prep data gradient at the sizes of the step: 16 real channels zero-padded to one K chunk of 32, the stage
    output's ReLU mask and the gradient that came back through the next stage's pool added in place."""
    _assert_tile(ops, n, h, w, 16, ci, tile)
    dy = bf(gen(n, 16, h, w, seed=124))
    wt = gen(16, ci, 3, 3, seed=125, scale=0.05)
    xin = torch.zeros(n, ci, h, w, requires_grad=True)
    F.conv2d(xin, bf(wt), None, padding=1).backward(dy)
    ref = xin.grad
    _, wd = ops.pack_conv3x3_weights(wt.to(DEV))
    dy_pad = torch.zeros(n, h, w, 32, dtype=torch.bfloat16, device=DEV)
    dy_pad[..., :16] = to_nhwc_bf16(dy)
    act = bf(F.relu(gen(n, ci, h, w, seed=126)))
    add = bf(gen(n, ci, h, w, seed=127))
    add_dev = to_nhwc_bf16(add)
    dx = ops.conv3x3_dgrad(dy_pad, wd, ci, 16, relu_src=to_nhwc_bf16(act), addend=add_dev, out=add_dev)
    ref2 = bf(bf(ref) * (act > 0).float() + add)
    got = from_nhwc(dx)
    tol = (2.0 ** -7) * ref2.abs() + 1e-5 * ref2.abs().max() + (2.0 ** -7) * ref.abs()
    assert ((got - ref2).abs() <= tol).all(), f"side dgrad {tile}: rel err {rel_err(got, ref2):.3e}"


def test_conv3x3_side_prep_fwd_large_map(ops):
    """side_prep forward at a five-frame stage-2 map: the 256-pixel x 16-channel tile, fp32 output."""
    n, h, w, ci = 5, 240, 427, 128
    assert ops.conv3x3_plan(n, h, w, ci, 16)["tile"] == (8, 32, 16)
    x = bf(gen(n, ci, h, w, seed=128))
    wt = gen(16, ci, 3, 3, seed=129, scale=math.sqrt(1.0 / (9 * ci)))
    b = gen(16, seed=130, scale=0.2)
    wf, _ = ops.pack_conv3x3_weights(wt.to(DEV))
    y = ops.conv3x3_fwd(to_nhwc_bf16(x), wf, b.to(DEV), ci, 16, relu=False, out_f32=True)
    assert rel_err(from_nhwc(y), F.conv2d(x, bf(wt), b, padding=1)) < 2e-5


def test_conv_first_fwd_persistent_loop(ops):
    """conv1_1 at 1x480x854: 1620 tiles on 1024 persistent workgroups, so the tile loop and its double-buffered staging
    iterate (every smaller case above runs one tile per workgroup)."""
    n, h, w = 1, 480, 854
    tiles, wgs = ops.conv3x3_first_plan(n, h, w)
    assert tiles > wgs == 1024
    x = gen(n, 3, h, w, seed=131, scale=60.0)
    wt = gen(64, 3, 3, 3, seed=132, scale=0.2)
    b = gen(64, seed=133, scale=0.5)
    y = ops.conv3x3_first_fwd(x.to(DEV), wt.to(DEV), b.to(DEV))
    assert_bf16_close(from_nhwc(y), F.relu(F.conv2d(x, wt, b, padding=1)), "conv_first_fwd 480x854")


def test_conv3x3_wgrad_five_frames(ops):
    """Weight gradient of a five-frame launch at a stage-3 map (what the step runs: 192 pixel splits, tiles of several
    images per split), against torch fp32 on the same bf16 operands; bit-reproducible."""
    n, h, w, ci, co = 5, 120, 214, 128, 256
    x = bf(gen(n, ci, h, w, seed=134))
    dy = bf(gen(n, co, h, w, seed=135))
    wt = torch.zeros(co, ci, 3, 3, requires_grad=True)
    b = torch.zeros(co, requires_grad=True)
    F.conv2d(x, wt, b, padding=1).backward(dy)
    xd, dyd = to_nhwc_bf16(x), to_nhwc_bf16(dy)
    dw, db = ops.conv3x3_wgrad(xd, dyd, ci, co)
    assert rel_err(dw.cpu(), wt.grad) < 5e-5, f"wgrad rel err {rel_err(dw.cpu(), wt.grad):.3e}"
    assert rel_err(db.cpu(), b.grad) < 5e-5
    dw2, db2 = ops.conv3x3_wgrad(xd, dyd, ci, co)
    assert torch.equal(dw2, dw) and torch.equal(db2, db)


def _wgrad_ref(x, dy, co, ci):
    """torch fp32 on the CPU, image by image, the images' gradients added in float64 (so that the reference's own fp32
    accumulation over 2 M pixels is not what the comparison measures)."""
    dw = torch.zeros(co, ci, 3, 3, dtype=torch.float64)
    db = torch.zeros(co, dtype=torch.float64)
    for i in range(x.shape[0]):
        wt = torch.zeros(co, ci, 3, 3, requires_grad=True)
        b = torch.zeros(co, requires_grad=True)
        F.conv2d(x[i:i + 1], wt, b, padding=1).backward(dy[i:i + 1])
        dw += wt.grad.double()
        db += b.grad.double()
    return dw, db


# The weight-gradient kernels ON THE SHAPES OF THE 480x854 STEP (five frames per launch): the backbone form on conv1_2's map
# (2 M pixels over 192 pixel splits, multi-image tiles, the largest launch of the step), the side_prep form (16 outputs in a
# 32-wide gradient image) on the largest and the smallest side map, conv1_1's 27-column form.  Each against torch fp32 on
# the same bf16 operands; fixed-order reductions, so a second launch is bit-identical.
@pytest.mark.parametrize("n,h,w,ci,co", [(5, 480, 854, 64, 64), (5, 240, 427, 128, 16), (5, 30, 54, 512, 16)])
def test_conv3x3_wgrad_step_shapes(ops, n, h, w, ci, co):
    x = bf(gen(n, ci, h, w, seed=140))
    dy = bf(gen(n, co, h, w, seed=141))
    dw_ref, db_ref = _wgrad_ref(x, dy, co, ci)
    cy = (co + 31) // 32 * 32
    dy_dev = torch.zeros(n, h, w, cy, dtype=torch.bfloat16, device=DEV)
    dy_dev[..., :co] = to_nhwc_bf16(dy)
    xd = to_nhwc_bf16(x)
    del x, dy
    dw, db = ops.conv3x3_wgrad(xd, dy_dev, ci, co)
    assert dw.shape == (co, ci, 3, 3)
    e_w, e_b = rel_err(dw.cpu().double(), dw_ref), rel_err(db.cpu().double(), db_ref)
    print(f"[wgrad {n}x{h}x{w} {ci}->{co}] rel-to-max err dw {e_w:.2e} db {e_b:.2e}")
    assert e_w < 5e-5 and e_b < 5e-5
    dw2, db2 = ops.conv3x3_wgrad(xd, dy_dev, ci, co)
    assert torch.equal(dw2, dw) and torch.equal(db2, db)
    # accumulate mode on top of existing gradients (what the training loops run)
    dw3, db3 = ops.conv3x3_wgrad(xd, dy_dev, ci, co, dw=dw.clone(), db=db.clone(), accumulate=True)
    assert torch.equal(dw3, 2 * dw) and torch.equal(db3, 2 * db)


def test_conv_first_wgrad_step_shape(ops):
    """conv1_1's weight gradient at five 480x854 frames (the frame rounded to bf16 inside the kernel, like every other
    weight gradient reads bf16 activations; fp32 accumulate)."""
    n, h, w = 5, 480, 854
    x = gen(n, 3, h, w, seed=142, scale=60.0)
    dy = bf(gen(n, 64, h, w, seed=143))
    dw_ref, db_ref = _wgrad_ref(bf(x), dy, 64, 3)
    dyd = to_nhwc_bf16(dy)
    del dy
    dw, db = ops.conv3x3_first_wgrad(x.to(DEV), dyd)
    e_w, e_b = rel_err(dw.cpu().double(), dw_ref), rel_err(db.cpu().double(), db_ref)
    print(f"[first wgrad {n}x{h}x{w}] rel-to-max err dw {e_w:.2e} db {e_b:.2e}")
    assert e_w < 5e-5 and e_b < 5e-5
    dw2, db2 = ops.conv3x3_first_wgrad(x.to(DEV), dyd)
    assert torch.equal(dw2, dw) and torch.equal(db2, db)


# ------------------------------------------------------------------------------------------ pool
@pytest.mark.parametrize("n,h,w,c", [(1, 48, 86, 64), (2, 61, 107, 64), (1, 1, 1, 64), (1, 2, 3, 128), (1, 7, 1, 64)])
def test_maxpool(ops, n, h, w, c):
    x = bf(F.relu(gen(n, c, h, w, seed=40)))  # ReLU output: many exact zeros and ties
    xr = x.clone().requires_grad_(True)
    ref = F.max_pool2d(xr, 2, 2, ceil_mode=True)
    y = ops.maxpool_fwd(to_nhwc_bf16(x))
    assert y.shape == (n, (h + 1) // 2, (w + 1) // 2, c)
    assert torch.equal(from_nhwc(y), ref.detach())  # selection: exact
    dy = bf(gen(*ref.shape, seed=41))
    ref.backward(dy)
    dx = ops.maxpool_bwd(to_nhwc_bf16(x), to_nhwc_bf16(dy), relu_mask=False)
    assert torch.equal(from_nhwc(dx), xr.grad)      # routing to the first maximum: exact
    dxm = ops.maxpool_bwd(to_nhwc_bf16(x), to_nhwc_bf16(dy), relu_mask=True)
    assert torch.equal(from_nhwc(dxm), xr.grad * (x > 0).float())


# ------------------------------------------------------------------------------------------ head
def _head_inputs(n, H, W, seed):
    sizes = []
    h, w = H, W
    for _ in range(4):
        h, w = (h + 1) // 2, (w + 1) // 2
        sizes.append((h, w))
    g = torch.Generator().manual_seed(seed)
    side = [torch.randn(n, 16, hh, ww, generator=g) for hh, ww in sizes]
    # general per-channel k x k filters (not only bilinear) to exercise the real contract
    up = []
    up1 = []
    for i in range(4):
        k = 4 << i
        wt = torch.zeros(16, 16, k, k)
        base = torch.from_numpy(O.bilinear_kernel(k)).float()
        for c in range(16):
            wt[c, c] = base * (1.0 + 0.05 * c) + 0.01 * torch.randn(k, k, generator=g)
        up.append(wt)
        up1.append((base * 0.9 + 0.01 * torch.randn(k, k, generator=g)).reshape(1, 1, k, k))
    dsn_w = torch.randn(4, 16, generator=g) * 0.3
    dsn_b = torch.randn(4, generator=g) * 0.1
    fuse_w = torch.randn(64, generator=g) * 0.2
    fuse_b = torch.randn(1, generator=g)
    return side, up, up1, dsn_w, dsn_b, fuse_w, fuse_b


def _head_ref(side, up, up1, dsn_w, dsn_b, fuse_w, fuse_b, H, W):
    sides, outs = [], []
    for i in range(4):
        f = 2 << i
        sides.append(O.center_crop(F.conv_transpose2d(side[i], up[i], stride=f), H, W))
        score = F.conv2d(side[i], dsn_w[i].view(1, 16, 1, 1), dsn_b[i:i + 1])
        outs.append(O.center_crop(F.conv_transpose2d(score, up1[i], stride=f), H, W))
    fused = F.conv2d(torch.cat(sides, 1), fuse_w.view(1, 64, 1, 1), fuse_b)
    return outs + [fused]


@pytest.mark.parametrize("n,h,w,co", [(1, 48, 86, 64), (2, 61, 107, 128), (1, 30, 54, 512), (3, 480, 854, 64)])
def test_relu_mask_as_bits(ops, n, h, w, co):
    """conv1_1's forward can write the ReLU mask of its output as one bit per element (fosvos_conv3x3_first_fwd_bits), and the
    data gradient of the next conv can take its mask that way (fosvos_conv3x3_dgrad_bits: 8 instead of 128 bytes per pixel at
    64 channels - conv1_2's data gradient at 480x854 is bound by HBM traffic): the bits ARE (y > 0), and the data gradient is
    bit for bit the one computed from y itself - in the fused epilogue and in the split-K epilogue (the small shapes)."""
    g = torch.Generator().manual_seed(400 + h)
    if co == 64:
        frame = torch.randn(n, 3, h, w, generator=g).to(DEV)
        w1 = (torch.randn(64, 3, 3, 3, generator=g) * 0.3).to(DEV)
        b1 = (torch.randn(64, generator=g) * 0.5).to(DEV)
        y, bits = ops.conv3x3_first_fwd(frame, w1, b1, want_bits=True)
        assert torch.equal(y, ops.conv3x3_first_fwd(frame, w1, b1))
    else:  # any other producer: the bits are taken on the host here
        y = torch.relu(torch.randn(n, h, w, co, generator=g)).to(torch.bfloat16).to(DEV)
        bits = None
    pos = (y.float() > 0).reshape(n, h, w, co // 8, 8).to(torch.uint8)
    want_bits = (pos * (2 ** torch.arange(8, device=DEV, dtype=torch.uint8))).sum(-1).to(torch.uint8)
    if bits is not None:
        assert torch.equal(bits, want_bits) and 0.2 < pos.float().mean().item() < 0.8
    bits = want_bits.contiguous()
    cout = 64
    wt = torch.randn(cout, co, 3, 3, generator=g) * 0.05
    _, wd = ops.pack_conv3x3_weights(wt.to(DEV))
    dy = torch.randn(n, h, w, cout, generator=g).to(torch.bfloat16).to(DEV)
    add = torch.randn(n, h, w, co, generator=g).to(torch.bfloat16).to(DEV)
    for addend in (None, add):
        a = ops.conv3x3_dgrad(dy, wd, co, cout, relu_src=y, addend=addend)
        b = ops.conv3x3_dgrad(dy, wd, co, cout, relu_bits=bits, addend=addend)
        assert torch.equal(a, b)


@pytest.mark.parametrize("n,h,w,ci,tile", [
    (1, 48, 86, 128, (8, 16, 64)), (2, 61, 107, 64, (8, 16, 64)), (1, 33, 47, 256, (8, 16, 64)), (3, 17, 16, 64, (8, 16, 64)),
    (5, 240, 427, 128, (16, 16, 64)), (5, 120, 214, 256, (8, 32, 64)), (5, 60, 107, 512, (16, 16, 64)),  # the step's launches
    (2, 61, 107, 32, None),  # no fused form for 32 channels: the library runs the two passes itself
])
def test_dgrad_with_fused_pool_backward(ops, n, h, w, ci, tile):
    """fosvos_conv3x3_dgrad_unpool: side_prep's data gradient on a stage output x with the backward of the stage's 2x2 ceil-mode
    max pool in its epilogue (src/networks/osvos_vgg.py:61-83: x feeds side_prep[i] AND stages[i+1]'s pool).  Bit for bit the
    two passes it replaces - fosvos_maxpool2x2_ceil_bwd(x, d_pooled, relu mask) then fosvos_conv3x3_dgrad(relu_src=x, addend=that) -
    on inputs full of ties (x takes few distinct values and many zeros: first maximum in scan order wins, a dead window gets
    nothing) and ragged last rows / columns; on all three tiles that have the fused epilogue, the three launches of the training
    step among them (tile asserted), and on a shape without one.  And against autograd on fp32 at one shape."""
    g = torch.Generator().manual_seed(700 + h + ci)
    co = 16
    x = (torch.relu(torch.randn(n, h, w, ci, generator=g)) * 2).round().div(2).to(torch.bfloat16).to(DEV)  # multiples of 0.5: ties
    assert 0.3 < (x == 0).float().mean().item() < 0.7
    dpool = torch.randn(n, (h + 1) // 2, (w + 1) // 2, ci, generator=g).to(torch.bfloat16).to(DEV)
    wt = torch.randn(co, ci, 3, 3, generator=g) * 0.1
    _, wd = ops.pack_conv3x3_weights(wt.to(DEV))
    dy = torch.zeros(n, h, w, 32, dtype=torch.bfloat16, device=DEV)
    dy[..., :co] = torch.randn(n, h, w, co, generator=g).to(torch.bfloat16).to(DEV)
    if tile is not None:
        plan = ops.conv3x3_plan(n, h, w, co, ci)
        assert plan["tile"] == tile and plan["k_splits"] == 1, plan
    routed = ops.maxpool_bwd(x, dpool, relu_mask=True)
    want = ops.conv3x3_dgrad(dy, wd, ci, co, relu_src=x, addend=routed)
    got = ops.conv3x3_dgrad_unpool(dy, wd, ci, co, x, dpool)
    assert torch.equal(got, want)
    out = torch.full_like(got, 7.0)  # into a given buffer
    assert ops.conv3x3_dgrad_unpool(dy, wd, ci, co, x, dpool, out=out) is out and torch.equal(out, want)
    with pytest.raises(ValueError):
        ops.conv3x3_dgrad_unpool(dy, wd, ci, co, x, dpool, out=x)
    if (n, h, w, ci) == (2, 61, 107, 64):  # autograd: both consumers of relu(pre), on the same bf16-rounded operands
        pre = x.float().permute(0, 3, 1, 2).cpu()
        pre = torch.where(pre > 0, pre, -torch.ones_like(pre)).requires_grad_(True)
        xr = torch.relu(pre)
        side = F.conv2d(xr, wt.to(torch.bfloat16).float(), padding=1)
        pooled = F.max_pool2d(xr, 2, 2, ceil_mode=True)
        (side * dy[..., :co].float().permute(0, 3, 1, 2).cpu()).sum().backward(retain_graph=True)
        g_side = pre.grad.clone(); pre.grad = None
        (pooled * dpool.float().permute(0, 3, 1, 2).cpu()).sum().backward()
        # (the kernel rounds the conv term to bf16 before the add, like the two passes)
        ref = g_side.to(torch.bfloat16).float() + pre.grad
        assert rel_err(got.float().permute(0, 3, 1, 2).cpu(), ref) < 6e-3


@pytest.mark.parametrize("n,H,W,uniform", [(1, 48, 86, 0b1111), (2, 61, 107, 0b0101), (1, 33, 47, 0b1010), (3, 17, 16, 0b1000)])
def test_head_channel_uniform_filters(ops, n, H, W, uniform):
    """filt_uniform: where a scale's 16 channel filters are identical (what interp_surgery writes and lr 0 keeps,
    src/layers/osvos_layers.py:70-81) the head kernels contract the channels before the upsampling.  Against the torch
    reference (conv_transpose2d + crop + cat + 1x1, autograd) and against the general kernels on the same inputs - forward
    with and without side outputs, backward with all five upstream gradients and with the fused one only - for masks that mix
    uniform and per-channel scales."""
    side, up, up1, dsn_w, dsn_b, fuse_w, fuse_b = _head_inputs(n, H, W, seed=52)
    for i in range(4):
        if (uniform >> i) & 1:  # the same k x k filter (channel 3's, noise and all) on the whole diagonal
            for c in range(16):
                up[i][c, c] = up[i][3, 3]
    leaves = [s.clone().requires_grad_(True) for s in side]
    dw_l, db_l = dsn_w.clone().requires_grad_(True), dsn_b.clone().requires_grad_(True)
    fw_l, fb_l = fuse_w.clone().requires_grad_(True), fuse_b.clone().requires_grad_(True)
    ref = _head_ref(leaves, up, up1, dw_l, db_l, fw_l, fb_l, H, W)
    dev = lambda t: t.contiguous().to(DEV)
    side_d = [dev(s.permute(0, 2, 3, 1)) for s in side]
    idx = torch.arange(16)
    filt = [dev(u[idx, idx].permute(1, 2, 0)) for u in up]
    filt1 = [dev(u[0, 0]) for u in up1]
    assert ops.filters_uniform_mask(filt) == uniform
    args = (side_d, filt, filt1, dev(dsn_w), dev(dsn_b), dev(fuse_w), dev(fuse_b), H, W, True)
    fused, so = ops.head_fwd(*args, filt_uniform=uniform)
    fused_g, so_g = ops.head_fwd(*args, filt_uniform=0)
    assert rel_err(fused.cpu(), ref[4].detach()) < 1e-5 and rel_err(fused, fused_g) < 2e-6
    for i in range(4):
        assert rel_err(so[i].cpu(), ref[i].detach()) < 1e-5 and rel_err(so[i], so_g[i]) < 2e-6, f"side_out {i}"
    fused_only, _ = ops.head_fwd(side_d, filt, None, None, None, dev(fuse_w), dev(fuse_b), H, W, False, filt_uniform=uniform)
    assert rel_err(fused_only, fused) < 1e-6
    g = [gen(n, 1, H, W, seed=64 + i) for i in range(5)]
    torch.autograd.backward(ref, g)
    bargs = (side_d, filt, filt1, dev(dsn_w), dev(fuse_w), dev(g[4]), [dev(t) for t in g[:4]], H, W)
    d_side, d_fw, d_fb, d_dw, d_db = ops.head_bwd(*bargs, filt_uniform=uniform)
    d_side_g, d_fw_g, d_fb_g, d_dw_g, d_db_g = ops.head_bwd(*bargs, filt_uniform=0)
    for i in range(4):
        got = d_side[i].float().cpu()
        assert torch.equal(got[..., 16:], torch.zeros_like(got[..., 16:]))
        assert_bf16_close(got[..., :16].permute(0, 3, 1, 2), leaves[i].grad, f"d_side {i}")
        assert rel_err(d_side[i].float(), d_side_g[i].float()) < 4e-3  # (two bf16 roundings of nearly equal fp32 sums)
    for got, gen_, want in ((d_fw, d_fw_g, fw_l.grad), (d_fb, d_fb_g, fb_l.grad), (d_dw, d_dw_g, dw_l.grad), (d_db, d_db_g, db_l.grad)):
        assert rel_err(got.cpu(), want) < 2e-5 and rel_err(got, gen_) < 1e-5
    d_side2, d_fw2, d_fb2, a, b = ops.head_bwd(side_d, filt, None, None, dev(fuse_w), dev(g[4]), None, H, W, filt_uniform=uniform)
    d_side2_g, d_fw2_g, _, _, _ = ops.head_bwd(side_d, filt, None, None, dev(fuse_w), dev(g[4]), None, H, W)
    assert a is None and b is None and rel_err(d_fw2, d_fw2_g) < 1e-5
    for i in range(4):
        assert rel_err(d_side2[i].float(), d_side2_g[i].float()) < 4e-3


@pytest.mark.parametrize("n,H,W", [(1, 48, 86), (2, 61, 107), (1, 33, 47), (1, 17, 16)])
def test_head_fwd_bwd(ops, n, H, W):
    side, up, up1, dsn_w, dsn_b, fuse_w, fuse_b = _head_inputs(n, H, W, seed=50)
    leaves = [s.clone().requires_grad_(True) for s in side]
    dw_l, db_l = dsn_w.clone().requires_grad_(True), dsn_b.clone().requires_grad_(True)
    fw_l, fb_l = fuse_w.clone().requires_grad_(True), fuse_b.clone().requires_grad_(True)
    ref = _head_ref(leaves, up, up1, dw_l, db_l, fw_l, fb_l, H, W)
    dev = lambda t: t.contiguous().to(DEV)
    side_d = [dev(s.permute(0, 2, 3, 1)) for s in side]
    idx = torch.arange(16)
    filt = [dev(u[idx, idx].permute(1, 2, 0)) for u in up]  # [k,k,16], channel fastest
    filt1 = [dev(u[0, 0]) for u in up1]
    fused, so = ops.head_fwd(side_d, filt, filt1, dev(dsn_w), dev(dsn_b), dev(fuse_w), dev(fuse_b), H, W, True)
    assert rel_err(fused.cpu(), ref[4].detach()) < 1e-5
    for i in range(4):
        assert rel_err(so[i].cpu(), ref[i].detach()) < 1e-5, f"side_out {i}"
    fused_only, none = ops.head_fwd(side_d, filt, None, None, None, dev(fuse_w), dev(fuse_b), H, W, False)
    assert none is None and rel_err(fused_only, fused) < 1e-6  # other template instance: FMA order may differ
    # backward: all five upstream gradients
    g = [gen(n, 1, H, W, seed=60 + i) for i in range(5)]
    torch.autograd.backward(ref, g)
    d_side, d_fw, d_fb, d_dw, d_db = ops.head_bwd(side_d, filt, filt1, dev(dsn_w), dev(fuse_w), dev(g[4]),
                                                  [dev(t) for t in g[:4]], H, W)
    for i in range(4):
        got = d_side[i].float().cpu()
        assert torch.equal(got[..., 16:], torch.zeros_like(got[..., 16:])), "padding channels must be zero"
        assert_bf16_close(got[..., :16].permute(0, 3, 1, 2), leaves[i].grad, f"d_side {i}")
    assert rel_err(d_fw.cpu(), fw_l.grad) < 2e-5
    assert rel_err(d_fb.cpu(), fb_l.grad) < 2e-5
    assert rel_err(d_dw.cpu(), dw_l.grad) < 2e-5
    assert rel_err(d_db.cpu(), db_l.grad) < 2e-5
    # backward: fused gradient only (the online objective)
    for t in leaves + [fw_l, fb_l]:
        t.grad = None
    ref2 = _head_ref(leaves, up, up1, dw_l, db_l, fw_l, fb_l, H, W)
    ref2[4].backward(g[4])
    d_side2, d_fw2, d_fb2, a, b = ops.head_bwd(side_d, filt, None, None, dev(fuse_w), dev(g[4]), None, H, W)
    assert a is None and b is None
    for i in range(4):
        assert_bf16_close(d_side2[i].float().cpu()[..., :16].permute(0, 3, 1, 2), leaves[i].grad, f"d_side(fused only) {i}")
    assert rel_err(d_fw2.cpu(), fw_l.grad) < 2e-5 and rel_err(d_fb2.cpu(), fb_l.grad) < 2e-5


# ------------------------------------------------------------------------------------------ loss
@pytest.mark.parametrize("tag", ["kat12", "allneg", "allpos", "extreme", "rand", "soft"])
def test_cbce_golden(ops, golden, tag):
    k = golden("kat.npz")
    x, y = torch.from_numpy(k[f"loss_{tag}_x"]), torch.from_numpy(k[f"loss_{tag}_y"])
    loss, grad = ops.cbce_loss(x.to(DEV), y.to(DEV), size_average=False)
    np.testing.assert_allclose(loss.item(), k[f"loss_{tag}_sum"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(grad.cpu().numpy(), k[f"loss_{tag}_grad"], rtol=1e-5, atol=1e-7)
    loss_a, grad_a = ops.cbce_loss(x.to(DEV), y.to(DEV), size_average=True, grad_scale=0.2)
    np.testing.assert_allclose(loss_a.item(), k[f"loss_{tag}_avg"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(grad_a.cpu().numpy(), 0.2 * k[f"loss_{tag}_grad"] / x.numel(), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("shape", [(1, 1, 480, 854), (16, 1, 61, 107), (1, 1, 1, 1), (1, 1, 3, 5)])
def test_cbce_sizes(ops, shape):
    x = gen(*shape, seed=70, scale=5.0)
    _, y = O.synthetic_frame(shape[0], shape[2], shape[3], seed=71)
    ref = O.cbce_loss(x.double(), y.double(), size_average=False)
    gref = O.cbce_loss_grad(x.double(), y.double(), size_average=False)
    loss, grad = ops.cbce_loss(x.to(DEV), y.to(DEV), size_average=False)
    assert abs(loss.item() - ref.item()) <= 2e-6 * abs(ref.item()) + 1e-6
    assert rel_err(grad.cpu().double(), gref) < 1e-6
    loss2, _ = ops.cbce_loss(x.to(DEV), y.to(DEV), size_average=False, want_grad=False)
    assert loss2.item() == loss.item()  # deterministic reduction


def test_cbce_announced_backward_seed():
    """The online loop announces the tensor it seeds the backward pass with (1 / nAveGrad): the loss kernel writes the
    gradient already multiplied and the backward pass of THAT tensor launches nothing; any other incoming gradient still
    gives the right answer.  Both the single-tensor loss and the per-frame loss of a batched pass."""
    from layers import osvos_layers as L
    x = gen(3, 1, 24, 36, seed=74).to(DEV)
    y = (gen(3, 1, 24, 36, seed=75) > 0.3).float().to(DEV)
    seed = torch.full((3,), 0.2, device=DEV)
    one = torch.ones((), device=DEV) / 5

    def run(fn, out_slice, seed_t, announce, incoming):
        xi = x[out_slice].clone().requires_grad_(True)
        loss = fn(xi, y[out_slice], size_average=False, backward_seed=(seed_t, 0.2) if announce else None)
        loss.backward(incoming)
        return xi.grad

    hits0 = L.seed_hits
    plain = run(L.class_balanced_cross_entropy_loss_frames, slice(0, 3), seed, False, seed)
    fast = run(L.class_balanced_cross_entropy_loss_frames, slice(0, 3), seed, True, seed)
    assert L.seed_hits == hits0 + 1
    assert rel_err(fast.cpu(), plain.cpu()) < 1e-6            # (scale folded into the class weights in fp64: <= 1 ulp apart)
    other = torch.tensor([0.5, 1.0, 0.25], device=DEV)
    got = run(L.class_balanced_cross_entropy_loss_frames, slice(0, 3), seed, True, other)
    want = run(L.class_balanced_cross_entropy_loss_frames, slice(0, 3), seed, False, other)
    assert L.seed_hits == hits0 + 1 and rel_err(got.cpu(), want.cpu()) < 1e-6
    plain1 = run(L.class_balanced_cross_entropy_loss, slice(0, 1), one, False, one)
    fast1 = run(L.class_balanced_cross_entropy_loss, slice(0, 1), one, True, one)
    assert L.seed_hits == hits0 + 2 and rel_err(fast1.cpu(), plain1.cpu()) < 1e-6
    assert torch.equal(fast1, fast[0:1])                       # the batched pass's frame 0 == the frame alone


def test_cbce_frames_staged_equals_one_call(ops):
    """The per-frame loss in three launches with other work in between (count the labels' classes in front of the forward
    pass, loss + gradient between the passes, loss values behind the backward pass: fosvos_cbce_loss_frames_parts) gives
    the bits of the one-call form - values and gradients - also through the autograd wrapper the online loop uses."""
    from layers import osvos_layers as L
    x = gen(3, 1, 48, 86, seed=76, scale=4.0).to(DEV)
    y = (gen(3, 1, 48, 86, seed=77) > 0.3).float().to(DEV)
    want_l, want_g = ops.cbce_loss_frames(x, y, size_average=False, grad_scale=0.2)
    staged = ops.CbceFramesStaged(y)
    junk = torch.randn(1 << 20, device=DEV).sum()          # (other kernels and allocations between the stages)
    got_l, got_g = staged.loss(x, size_average=False, grad_scale=0.2)
    junk = junk + torch.randn(1 << 20, device=DEV).sum()
    assert staged.finish() is got_l
    assert torch.equal(got_l, want_l) and torch.equal(got_g, want_g)
    # autograd path
    seed = torch.full((3,), 0.2, device=DEV)
    xa = x.clone().requires_grad_(True)
    st = L.stage_frames_loss(y)
    assert st is not None
    loss = L.class_balanced_cross_entropy_loss_frames(xa, y, size_average=False, backward_seed=(seed, 0.2), staged=st)
    loss.backward(seed)
    st.finish()
    assert torch.equal(loss.detach(), want_l) and torch.equal(xa.grad, want_g)
    with pytest.raises(ValueError):
        L.class_balanced_cross_entropy_loss_frames(xa, y.clone(), size_average=False, staged=st)
    assert L.stage_frames_loss(y[:, :, :3, :5].contiguous()) is None  # 15 elements per frame: the one-call path


# ------------------------------------------------------------------------------------------ SGD
def test_fused_sgd_zero_grad_flag():
    """step(zero_grad=True) = step() then zero_grad(set_to_none=False), in one pass over the gradients - also for a subset."""
    from fosvos_hip.sgd import FusedSGD
    g = torch.Generator().manual_seed(81)
    shapes = [(64, 3, 3, 3), (5,), (33, 7)]
    pa = [torch.randn(s, generator=g).to(DEV).requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa = FusedSGD(pa, lr=1e-2, momentum=0.9, weight_decay=1e-3)
    ob = FusedSGD(pb, lr=1e-2, momentum=0.9, weight_decay=1e-3)
    for step in range(3):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g).to(DEV)
            a.grad = gr.clone() if a.grad is None else a.grad.copy_(gr)
            b.grad = gr.clone()
        keep = [a.grad for a in pa]
        if step == 1:  # a subset: the others keep their gradients
            oa.step(only=pa[:2], tag="part", zero_grad=True)
            ob.step(only=pb[:2], tag="part")
            assert not pa[0].grad.any() and not pa[1].grad.any() and pa[2].grad.any()
            oa.step(only=pa[2:], tag="rest", zero_grad=True)
            ob.step(only=pb[2:], tag="rest")
        else:
            oa.step(zero_grad=True)
            ob.step()
        for a, b, k in zip(pa, pb, keep):
            assert torch.equal(a.detach(), b.detach())
            assert a.grad is k and not a.grad.any()            # same tensor, now zeros


def test_fused_sgd_matches_torch():
    from fosvos_hip.sgd import FusedSGD
    shapes = [(64, 3, 3, 3), (64,), (128, 64, 3, 3), (1, 64, 1, 1), (1,), (7, 5)]
    g = torch.Generator().manual_seed(80)
    p_ref = [torch.randn(s, generator=g).requires_grad_(True) for s in shapes]
    p_hip = [p.detach().clone().to(DEV).requires_grad_(True) for p in p_ref]

    def groups(ps):
        return [{"params": ps[:2], "weight_decay": 2e-4}, {"params": ps[2:4], "lr": 2e-3},
                {"params": ps[4:5], "lr": 0.0}, {"params": ps[5:], "lr": 1e-4, "weight_decay": 1e-2}]

    o_ref = torch.optim.SGD(groups(p_ref), lr=1e-3, momentum=0.9)
    o_hip = FusedSGD(groups(p_hip), lr=1e-3, momentum=0.9)
    for step in range(4):
        for pr, ph in zip(p_ref, p_hip):
            gr = torch.randn(pr.shape, generator=g)
            pr.grad = gr.clone()
            ph.grad = gr.clone().to(DEV)
        if step == 2:  # a parameter without gradient is skipped
            p_ref[5].grad = None
            p_hip[5].grad = None
        o_ref.step()
        o_hip.step()
        for pr, ph in zip(p_ref, p_hip):
            np.testing.assert_allclose(ph.detach().cpu().numpy(), pr.detach().numpy(), rtol=2e-6, atol=1e-7)
