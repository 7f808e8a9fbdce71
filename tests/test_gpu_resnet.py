"""OSVOS_RESNET inference on the HIP kernels (csrc/resnet.hip) against the CPU oracle
(oracle/osvos_resnet_ref.py), op by op and end to end, through the C ABI.

Tolerances: activations and packed weights are bf16 (8 mantissa bits), accumulation is fp32.  Op tests feed both sides
the SAME bf16-rounded operands, so what remains is summation order and the final bf16 store: 2^-8 relative to the
output scale.  Network tests compare with the pure fp32 oracle: logits within 3 % of the oracle's logit range, the
thresholded mask identical wherever the oracle's |logit| exceeds that tolerance.
"""
import os
import sys

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _nhwc_pad8(x_nchw):
    n, c, h, w = x_nchw.shape
    cp = (c + 7) // 8 * 8
    out = torch.zeros((n, h, w, cp), dtype=torch.bfloat16)
    out[..., :c] = x_nchw.permute(0, 2, 3, 1).to(torch.bfloat16)
    return out


def _bn_params(c, g):
    return (0.7 + 0.6 * torch.rand(c, generator=g), 0.1 * torch.randn(c, generator=g), 0.2 * torch.randn(c, generator=g),
            0.5 + torch.rand(c, generator=g), 1e-5)


@pytest.mark.parametrize("ci,co,k,stride,relu,with_bn,with_add,f32", [
    (8, 8, 3, 1, True, True, False, False),
    (13, 21, 3, 1, True, True, True, False),      # channel counts that are no multiple of anything
    (16, 16, 3, 2, True, True, False, False),
    (64, 72, 3, 1, False, True, True, False),     # 72 -> five blocks of 16
    (32, 64, 3, 2, True, True, False, False),
    (24, 40, 1, 2, False, True, False, False),    # the downsample branch
    (40, 160, 1, 1, True, True, True, False),     # Bottleneck's widening 1x1 + residual
    (128, 128, 3, 1, True, True, True, False),
    (59, 16, 3, 1, False, False, False, True),    # side_prep: conv bias, no BatchNorm, fp32 NHWC out
])
def test_conv2d_matches_torch(ci, co, k, stride, relu, with_bn, with_add, f32):
    from fosvos_hip import ops
    g = torch.Generator().manual_seed(ci * 131 + co * 7 + k + stride)
    n, h, w = 2, 19, 27
    x = _bf(torch.randn(n, ci, h, w, generator=g))
    wt = torch.randn(co, ci, k, k, generator=g) * (2.0 / (ci * k * k)) ** 0.5
    bn = _bn_params(co, g) if with_bn else None
    cbias = None if with_bn else 0.1 * torch.randn(co, generator=g)
    packed, bias = ops.pack_conv2d_bn(wt.to(DEV), None if cbias is None else cbias.to(DEV),
                                      None if bn is None else tuple(t.to(DEV) for t in bn[:4]) + (bn[4],))
    if bn is not None:
        s = bn[0] / torch.sqrt(bn[3] + bn[4])
        w_ref, b_ref = _bf(wt * s.view(-1, 1, 1, 1)), bn[1] - bn[2] * s
    else:
        w_ref, b_ref = _bf(wt), cbias
    ref = F.conv2d(x, w_ref, b_ref, stride=stride, padding=k // 2)
    add = None
    if with_add:
        add = _bf(torch.randn(ref.shape, generator=g))
        ref = ref + add
    if relu:
        ref = F.relu(ref)
    y = ops.conv2d_fwd(_nhwc_pad8(x).to(DEV), packed, bias, ci, co, k, stride, relu,
                       None if add is None else _nhwc_pad8(add).to(DEV), out_f32=f32)
    torch.cuda.synchronize()
    assert y.dtype == (torch.float32 if f32 else torch.bfloat16) and y.shape[3] == (co + 7) // 8 * 8
    got = y.float().cpu()
    assert torch.count_nonzero(got[..., co:]) == 0                    # padded channels are exact zeros
    got = got[..., :co].permute(0, 3, 1, 2)
    tol = (2.0 ** -8 if not f32 else 2.0 ** -14) * max(ref.abs().max().item(), 1.0) + 1e-5
    assert (got - ref).abs().max().item() <= tol


@pytest.mark.parametrize("ci,co,hw,relu,with_add", [
    (64, 64, (40, 56), True, True),       # BasicBlock conv2 + residual
    (64, 64, (40, 56), True, False),      # BasicBlock conv1
    (128, 128, (17, 30), True, True),     # small map: split-K epilogue kernel
    (32, 64, (33, 47), False, True),
    (256, 256, (9, 15), True, True),
    (32, 32, (300, 260), True, True),     # the 32-wide tile (thinned nets), large-map variant
    (32, 32, (33, 47), True, True),       # ... small-map variant
    (64, 32, (20, 31), False, False),
])
def test_mfma_residual_conv_matches_torch(ci, co, hw, relu, with_add):
    """The MFMA implicit GEMM in its residual form (fosvos_conv3x3_fwd_add) with BatchNorm folded by fosvos_fold_conv_bn:
    relu(conv + bias + addend).  The conv result is rounded to bf16 before the add (the tile is staged in bf16), hence
    2 x 2^-8 of the output scale."""
    from fosvos_hip import ops
    g = torch.Generator().manual_seed(ci + co + hw[0])
    n, (h, w) = 2, hw
    x = _bf(torch.randn(n, ci, h, w, generator=g))
    wt = torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (ci * 9)) ** 0.5
    bn = _bn_params(co, g)
    folded, bias = ops.fold_conv_bn(wt.to(DEV), None, tuple(t.to(DEV) for t in bn[:4]) + (bn[4],))
    s = bn[0] / torch.sqrt(bn[3] + bn[4])
    assert torch.allclose(folded.cpu(), wt * s.view(-1, 1, 1, 1), rtol=1e-6, atol=1e-7)
    assert torch.allclose(bias.cpu(), bn[1] - bn[2] * s, rtol=1e-6, atol=1e-6)
    packed, none = ops.pack_conv3x3_weights(folded, want_fwd=True, want_dgrad=False)
    assert none is None
    ref = F.conv2d(x, _bf(folded.cpu()), bias.cpu(), padding=1)
    add = None
    if with_add:
        add = _bf(torch.randn(ref.shape, generator=g))
        ref = ref + add
    if relu:
        ref = F.relu(ref)
    y = ops.conv3x3_fwd_add(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV), packed, bias, ci, co, relu,
                            None if add is None else add.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV))
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() <= 2.0 ** -7 * max(ref.abs().max().item(), 1.0) + 1e-5


@pytest.mark.parametrize("ci,co,hw", [(64, 128, (40, 56)), (32, 64, (33, 47)), (128, 256, (17, 31)), (256, 512, (9, 14)),
                                      (32, 32, (41, 37))])
def test_mfma_stride2_conv_matches_torch(ci, co, hw):
    """fosvos_conv3x3_s2_fwd: stride-2 conv as the stride-1 MFMA kernel with a subsampling store (odd sizes, and maps
    small enough for the split-K epilogue)."""
    from fosvos_hip import ops
    g = torch.Generator().manual_seed(ci + co)
    n, (h, w) = 2, hw
    x = _bf(torch.randn(n, ci, h, w, generator=g))
    wt = torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (ci * 9)) ** 0.5
    bias = 0.1 * torch.randn(co, generator=g)
    packed, _ = ops.pack_conv3x3_weights(wt.to(DEV), want_fwd=True, want_dgrad=False)
    ref = F.relu(F.conv2d(x, _bf(wt), bias, stride=2, padding=1))
    y = ops.conv3x3_s2_fwd(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV), packed, bias.to(DEV), ci, co, True)
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 2.0 ** -8 * max(ref.abs().max().item(), 1.0) + 1e-5


def test_first_conv_and_pool_match_torch():
    """Both forms of the 7x7 stride-2 layer: bf16 MFMA (the default; the reference gets the same bf16-rounded frame and
    folded weights) and fp32 vector-ALU math (FOSVOS_CONV_FP32_MATH)."""
    from fosvos_hip import ops
    g = torch.Generator().manual_seed(3)
    for co, (h, w) in ((16, (37, 53)), (64, (32, 48)), (21, (30, 31)), (8, (64, 80)), (40, (19, 130)), (32, (45, 132))):
        x = 60.0 * torch.randn(2, 3, h, w, generator=g)
        wt = torch.randn(co, 3, 7, 7, generator=g) * (2.0 / 147) ** 0.5 / 60.0
        bn = _bn_params(co, g)
        packed, bias = ops.pack_conv7x7_bn(wt.to(DEV), tuple(t.to(DEV) for t in bn[:4]) + (bn[4],))
        s = bn[0] / torch.sqrt(bn[3] + bn[4])
        for fp32_math in (False, True):
            if fp32_math:
                ref = F.relu(F.conv2d(x, wt * s.view(-1, 1, 1, 1), bn[1] - bn[2] * s, stride=2, padding=3))
            else:
                ref = F.relu(F.conv2d(_bf(x), _bf(wt * s.view(-1, 1, 1, 1)), bn[1] - bn[2] * s, stride=2, padding=3))
            y = ops.conv7x7s2_first_fwd(x.to(DEV), packed, bias, co, relu=True, fp32_math=fp32_math)
            p = ops.maxpool3x3s2_fwd(y)
            torch.cuda.synchronize()
            got = y.float().cpu()
            assert torch.count_nonzero(got[..., co:]) == 0
            got = got[..., :co].permute(0, 3, 1, 2)
            assert got.shape == ref.shape
            assert (got - ref).abs().max().item() <= 2.0 ** -8 * max(ref.abs().max().item(), 1.0) + 1e-5, (co, fp32_math)
            # the pool works on what the conv stored: bit-exact against torch on the same bf16 values
            want = F.max_pool2d(got, kernel_size=3, stride=2, padding=1)
            assert torch.equal(p.float().cpu()[..., :co].permute(0, 3, 1, 2), want)


def test_fused_first_conv_and_pool_equal_the_two_kernels():
    """fosvos_conv7x7s2_pool_first_fwd (the conv map stays in LDS) against conv kernel + pool kernel: same bits."""
    from fosvos_hip import ops
    g = torch.Generator().manual_seed(13)
    for co, (h, w) in ((16, (37, 53)), (32, (128, 250)), (64, (70, 64)), (21, (30, 31)), (8, (16, 20)), (40, (19, 130)),
                       (32, (270, 483)), (32, (135, 244)), (16, (33, 4)), (32, (61, 1920))):   # widths % 4 == 0: 16-byte loads
        x = 60.0 * torch.randn(2, 3, h, w, generator=g)
        wt = torch.randn(co, 3, 7, 7, generator=g) * (2.0 / 147) ** 0.5 / 60.0
        bn = _bn_params(co, g)
        packed, bias = ops.pack_conv7x7_bn(wt.to(DEV), tuple(t.to(DEV) for t in bn[:4]) + (bn[4],))
        want = ops.maxpool3x3s2_fwd(ops.conv7x7s2_first_fwd(x.to(DEV), packed, bias, co, relu=True))
        got = ops.conv7x7s2_pool_first_fwd(x.to(DEV), packed, bias, co)
        torch.cuda.synchronize()
        assert got.shape == want.shape and torch.equal(got, want), (co, h, w)


@pytest.mark.parametrize("h,w", [(64, 96), (70, 101), (33, 47)])
def test_deconv_head_matches_torch(h, w):
    from fosvos_hip import ops
    from oracle import osvos_resnet_ref as R
    g = torch.Generator().manual_seed(h * 1000 + w)
    n = 2
    sizes = []
    hh, ww = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    for _ in range(4):
        hh, ww = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
        sizes.append((hh, ww))
    side = [torch.randn(n, 16, a, b, generator=g) for a, b in sizes]
    sd = R.make_state_dict(18, 3, seed=5)
    fuse_w, fuse_b = sd["layer_fuse.weight"], sd["layer_fuse.bias"]
    ups, outs_ref = [], []
    for s in range(4):
        f = 4 << s
        ups.append(R.center_crop(F.conv_transpose2d(side[s], sd["upscale_side_prep.%d.weight" % s], stride=f), h, w))
        dsn = F.conv2d(side[s], sd["score_dsn.%d.weight" % s], sd["score_dsn.%d.bias" % s])
        outs_ref.append(R.center_crop(F.conv_transpose2d(dsn, sd["upscale_score_dsn.%d.weight" % s], stride=f), h, w))
    fused_ref = F.conv2d(torch.cat(ups, 1), fuse_w, fuse_b)
    filt = [torch.einsum("o,iokl->kli", fuse_w[0, 16 * s:16 * s + 16, 0, 0], sd["upscale_side_prep.%d.weight" % s]).contiguous().to(DEV)
            for s in range(4)]
    filt1 = [sd["upscale_score_dsn.%d.weight" % s][0, 0].contiguous().to(DEV) for s in range(4)]
    dsn_w = torch.cat([sd["score_dsn.%d.weight" % s].reshape(1, 16) for s in range(4)]).to(DEV)
    dsn_b = torch.cat([sd["score_dsn.%d.bias" % s] for s in range(4)]).to(DEV)
    side_nhwc = [t.permute(0, 2, 3, 1).contiguous().to(DEV) for t in side]
    fused, outs = ops.deconv_head_fwd(side_nhwc, [4, 8, 16, 32], filt, filt1, dsn_w, dsn_b, fuse_b.to(DEV), h, w, True)
    fused2, none = ops.deconv_head_fwd(side_nhwc, [4, 8, 16, 32], filt, None, None, None, fuse_b.to(DEV), h, w, False)
    torch.cuda.synchronize()
    assert none is None and torch.equal(fused, fused2)
    assert (fused.cpu() - fused_ref).abs().max().item() <= 1e-4 * max(1.0, fused_ref.abs().max().item())
    for a, b in zip(outs, outs_ref):
        assert (a.cpu() - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item())


def _check_net(net, sd, x, frac=3e-2):
    from oracle import osvos_resnet_ref as R
    net = net.to(DEV).eval()
    outs = net(x.to(DEV))
    torch.cuda.synchronize()
    ref = R.forward(sd, x)
    assert len(outs) == 5
    for i, (a, b) in enumerate(zip(outs, ref)):
        assert a.shape == b.shape and a.dtype == torch.float32
        tol = frac * b.abs().max().item()
        err = (a.cpu() - b).abs().max().item()
        assert err <= tol, f"output {i}: max |diff| {err:.4g} > {tol:.4g} ({frac:.0%} of the logit range)"
        sure = b.abs() > tol
        assert torch.equal((a.cpu() > 0)[sure], (b > 0)[sure]), f"output {i}: mask differs where the oracle is confident"
    return outs, ref


@pytest.mark.parametrize("version,e,hw", [(18, 0, (64, 96)), (18, 2, (97, 130)), (18, 3, (120, 200)), (34, 2, (70, 101)),
                                          (18, 2, (16, 20)), (18, 1, (33, 17))])   # ... down to 1 x 1 stage-4 maps
def test_network_matches_oracle(version, e, hw):
    from networks.osvos_resnet import OSVOS_RESNET
    from oracle import osvos_resnet_ref as R
    sd = R.make_state_dict(version, e, seed=version + e)
    net = OSVOS_RESNET(pretrained=False, version=version, scale_down_exponent=e)
    net.load_state_dict(sd)
    x = 50.0 * torch.randn(2, 3, *hw, generator=torch.Generator().manual_seed(9))
    _check_net(net, sd, x)


def test_wide_net_on_the_vector_alu_path_only(monkeypatch):
    """FOSVOS_RESNET_MFMA=0: the 64..512-channel layers of the full-width net on the direct kernel (the default routes
    them to the MFMA implicit GEMM)."""
    from networks.osvos_resnet import OSVOS_RESNET
    from oracle import osvos_resnet_ref as R
    monkeypatch.setenv("FOSVOS_RESNET_MFMA", "0")
    sd = R.make_state_dict(18, 0, seed=21)
    net = OSVOS_RESNET(pretrained=False, version=18, scale_down_exponent=0)
    net.load_state_dict(sd)
    x = 50.0 * torch.randn(1, 3, 64, 96, generator=torch.Generator().manual_seed(9))
    outs, _ = _check_net(net, sd, x)
    assert all(b.convs[0].kind == 0 for st in net._plan.stages for b in st)
    monkeypatch.setenv("FOSVOS_RESNET_MFMA", "1")
    outs2 = net(x.to(DEV))                                  # same weights, plan rebuilt with the MFMA layers
    assert any(b.convs[0].kind == 1 for st in net._plan.stages for b in st)
    ref = R.forward(sd, x)[-1]
    assert (outs2[-1].cpu() - ref).abs().max().item() <= 3e-2 * ref.abs().max().item()


def test_native_loop_equals_the_op_by_op_loop():
    from fosvos_hip import resnet_engine
    from networks.osvos_resnet import OSVOS_RESNET
    from oracle import osvos_resnet_ref as R
    for version, e, hw in ((18, 1, (65, 99)), (34, 3, (128, 160)), (18, 0, (64, 80))):
        net = OSVOS_RESNET(pretrained=False, version=version, scale_down_exponent=e)
        net.load_state_dict(R.make_state_dict(version, e, seed=3))
        net = net.to(DEV).eval()
        x = (50.0 * torch.randn(2, 3, *hw, generator=torch.Generator().manual_seed(5))).to(DEV)
        a = net(x)
        b = resnet_engine.forward_ops(net, net._plan, x)
        a2 = net(x)                                          # the arena is reused
        os.environ["FOSVOS_RESNET_AUX"] = "1"                # side_prep / downsample convs on the auxiliary stream
        try:
            a3 = net(x)
            a4 = net(x)
        finally:
            del os.environ["FOSVOS_RESNET_AUX"]
        torch.cuda.synchronize()
        for u, v, u2, u3, u4 in zip(a, b, a2, a3, a4):
            assert torch.equal(u, v) and torch.equal(u, u2) and torch.equal(u, u3) and torch.equal(u, u4)


def test_weight_update_repacks_and_batch_of_one_equals_batch_rows():
    from networks.osvos_resnet import OSVOS_RESNET
    from oracle import osvos_resnet_ref as R
    sd = R.make_state_dict(18, 2, seed=4)
    net = OSVOS_RESNET(pretrained=False, scale_down_exponent=2)
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    x = 50.0 * torch.randn(2, 3, 64, 80, generator=torch.Generator().manual_seed(1))
    both = net(x.to(DEV))[-1]
    one = net(x[1:].to(DEV))[-1]
    assert torch.equal(both[1:], one)                       # no cross-sample state, deterministic kernels
    with torch.no_grad():
        net.layer_stages[2][0].bn1.running_var.mul_(4.0)    # a buffer write must invalidate the folded images
    sd2 = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    _check_net(net, sd2, x)
    assert not torch.equal(net(x.to(DEV))[-1], both)


def test_pruned_block_with_odd_channel_counts():
    """What src/prune.py:297-481 produces: filters removed from a block's first conv (and the matching BatchNorm
    entries and input planes of the second conv), blocks rebuilt as BasicBlockDummy."""
    from networks.osvos_resnet import OSVOS_RESNET, BasicBlockDummy
    from oracle import osvos_resnet_ref as R
    sd = R.make_state_dict(18, 1, seed=11)
    net = OSVOS_RESNET(pretrained=False, scale_down_exponent=1)
    net.load_state_dict(sd)
    blk = net.layer_stages[1][0]
    keep = torch.tensor([i for i in range(blk.conv1.out_channels) if i not in (3, 17, 18, 40, 63)])
    conv1 = nn.Conv2d(blk.conv1.in_channels, len(keep), 3, stride=blk.conv1.stride, padding=1, bias=False)
    conv1.weight.data = blk.conv1.weight.data[keep].clone()
    bn1 = nn.BatchNorm2d(len(keep))
    for name in ("weight", "bias", "running_mean", "running_var"):
        getattr(bn1, name).data = getattr(blk.bn1, name).data[keep].clone()
    conv2 = nn.Conv2d(len(keep), blk.conv2.out_channels, 3, padding=1, bias=False)
    conv2.weight.data = blk.conv2.weight.data[:, keep].clone()
    net.layer_stages[1][0] = BasicBlockDummy(conv1, bn1, blk.relu, conv2, blk.bn2, blk.downsample, blk.stride)
    sd2 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    assert sd2["layer_stages.1.0.conv1.weight"].shape[0] == 59
    x = 50.0 * torch.randn(1, 3, 72, 104, generator=torch.Generator().manual_seed(2))
    _check_net(net, sd2, x)


def test_bottleneck_trunk_runs_once_side_prep_fits():
    """The reference sizes side_prep for BasicBlock trunks, so its Bottleneck versions fail at the first side branch;
    the same net with side_prep widened to the stage outputs runs (1x1 convs, 4x expansion, 1x1 downsample)."""
    from networks.osvos_resnet import OSVOS_RESNET
    from oracle import osvos_resnet_ref as R
    net = OSVOS_RESNET(pretrained=False, version=50, scale_down_exponent=3)
    x = 50.0 * torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(6))
    with pytest.raises(RuntimeError, match="side_prep.*expects"):
        net.to(DEV).eval()(x.to(DEV))
    wide = [4 * c for c in (8, 16, 32, 64)]
    sd = R.make_state_dict(50, 3, seed=8, side_channels=wide)
    net = OSVOS_RESNET(pretrained=False, version=50, scale_down_exponent=3)
    for i, c in enumerate(wide):
        net.side_prep[i] = nn.Conv2d(c, 16, kernel_size=3, padding=1)
    net.load_state_dict(sd)
    _check_net(net, sd, x)


def test_full_size_frame_properties():
    """BASELINE configs[4] size (1920x1080), where the oracle takes too long for a unit test: the native loop equals the
    op-by-op loop bit for bit, repeated calls are deterministic, a batch row equals the same frame alone, and zero
    weights in the head give the bias everywhere."""
    from fosvos_hip import resnet_engine
    from networks.osvos_resnet import OSVOS_RESNET
    from oracle import osvos_resnet_ref as R
    net = OSVOS_RESNET(pretrained=False, scale_down_exponent=2)
    net.load_state_dict(R.make_state_dict(18, 2, seed=12))
    net = net.to(DEV).eval()
    x = (50.0 * torch.randn(2, 3, 1080, 1920, generator=torch.Generator().manual_seed(7))).to(DEV)
    a = net(x)
    b = resnet_engine.forward_ops(net, net._plan, x)
    c = net(x[1:])
    torch.cuda.synchronize()
    assert all(t.shape == (2, 1, 1080, 1920) and torch.isfinite(t).all() for t in a)
    for u, v, w in zip(a, b, c):
        assert torch.equal(u, v) and torch.equal(u[1:], w)
    net.compute_side_outputs = False                       # fused map only: same bits, empty placeholders for the rest
    d = net(x)
    assert torch.equal(d[-1], a[-1]) and all(t.numel() == 0 for t in d[:4])
    net.compute_side_outputs = True
    with torch.no_grad():
        net.layer_fuse.weight.zero_()
        net.layer_fuse.bias.fill_(0.25)
    fused = net(x)[-1]
    assert torch.all(fused == 0.25)


def test_loud_failures():
    from networks.osvos_resnet import OSVOS_RESNET
    net = OSVOS_RESNET(pretrained=False, scale_down_exponent=3).to(DEV)
    x = torch.randn(1, 3, 64, 64)
    with pytest.raises(RuntimeError, match="eval"):
        net.train()(x.to(DEV))
    with pytest.raises(RuntimeError, match="GPU"):
        net.eval()(x)
    with pytest.raises(RuntimeError, match="torchvision"):
        OSVOS_RESNET(pretrained=True)
    with pytest.raises(Exception, match="Invalid version"):
        OSVOS_RESNET(pretrained=False, version=19)
    with pytest.raises(RuntimeError, match="parameters only"):
        net.layer_stages[0][0](x)
