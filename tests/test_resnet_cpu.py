"""SURVEY §8 (f4), CPU side: OSVOS_RESNET's module surface against the reference's module tree
(src/networks/osvos_resnet.py:15-150), the oracle's own consistency (BatchNorm folding and the head contraction the
HIP path relies on are identities of the fp32 restatement), the host-side size queries of the C ABI, loud failures."""
import ctypes
import os
import pickle
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))

from oracle import osvos_resnet_ref as R  # noqa: E402


@pytest.mark.parametrize("version,e", [(18, 0), (18, 3), (34, 2), (50, 1), (101, 3), (152, 2)])
def test_state_dict_keys_follow_the_reference_module_tree(version, e):
    from networks.osvos_resnet import OSVOS_RESNET
    net = OSVOS_RESNET(pretrained=False, version=version, scale_down_exponent=e)
    spec = R.state_dict_spec(version, e)
    sd = net.state_dict()
    assert list(sd.keys()) == list(spec.keys())
    assert all(tuple(sd[k].shape) == spec[k] for k in spec)
    # names the reference's own code relies on (src/prune.py:297-481 walks layer_stages[i][j].conv1/bn1/conv2/bn2/downsample)
    assert "layer_base.0.weight" in sd and "layer_stages.3.0.downsample.0.weight" in sd
    assert "layer_stages.0.0.downsample.0.weight" not in sd or version >= 50   # BasicBlock stage 1 keeps the identity
    assert sd["layer_base.0.weight"].shape[0] == 64 // 2 ** e and sd["side_prep.3.weight"].shape[1] == 512 // 2 ** e
    assert sd["upscale_side_prep.3.weight"].shape == (16, 16, 64, 64) and sd["layer_fuse.weight"].shape == (1, 64, 1, 1)


def test_initialisation_and_pickle():
    from networks.osvos_resnet import OSVOS_RESNET, BasicBlock
    net = OSVOS_RESNET(pretrained=False, scale_down_exponent=1)
    assert isinstance(net.layer_stages[0][0], BasicBlock) and net.layer_stages[1][0].stride == 2
    assert abs(net.layer_stages[2][1].conv2.weight.std().item() - 1e-3) < 1e-4           # N(0, 1e-3) convs
    bn = net.layer_stages[2][1].bn2
    assert torch.all(bn.weight == 1) and torch.all(bn.bias == 0)
    for i in range(4):
        k = 8 << i
        w = net.upscale_side_prep[i].weight.data
        assert w.shape == (16, 16, k, k) and net.upscale_side_prep[i].stride == (k // 2, k // 2)
        filt = torch.from_numpy(R.bilinear_kernel(k))
        assert torch.allclose(w[3, 3], filt) and w[3, 4].abs().max() == 0                # diagonal bilinear filters
        assert torch.allclose(net.upscale_score_dsn[i].weight.data[0, 0], filt)
    assert net.side_prep[0].bias.abs().max() == 0
    clone = pickle.loads(pickle.dumps(net))                                               # no device caches inside
    assert list(clone.state_dict().keys()) == list(net.state_dict().keys()) and clone._plan.signature is None
    with pytest.raises(Exception, match="Invalid version"):
        OSVOS_RESNET(pretrained=False, version=20)
    with pytest.raises(RuntimeError, match="torchvision"):
        OSVOS_RESNET(pretrained=True)


def test_no_cpu_fallback():
    from networks.osvos_resnet import OSVOS_RESNET
    net = OSVOS_RESNET(pretrained=False, scale_down_exponent=3)
    x = torch.zeros(1, 3, 64, 64)
    with pytest.raises(RuntimeError, match="eval"):
        net(x)                                            # modules start in training mode, as in torch
    with pytest.raises(RuntimeError, match="GPU"):
        net.eval()(x)
    with pytest.raises(RuntimeError, match="parameters only"):
        net.layer_stages[0][0](x)


def test_oracle_batchnorm_folding_is_an_identity():
    """What the pack kernel does: w' = w s, b' = beta - mean s with s = gamma / sqrt(var + eps)."""
    sd = R.make_state_dict(18, 2, seed=3)
    x = 50.0 * torch.randn(1, 3, 70, 90, generator=torch.Generator().manual_seed(1))
    want = R.trunk(sd, x)
    folded = dict(sd)
    bias = {}
    for k in list(sd):
        if not k.endswith(".running_var"):
            continue
        bn = k[:-len(".running_var")]
        conv = {"layer_base.1": "layer_base.0"}.get(bn) or bn.replace(".bn", ".conv").replace("downsample.1", "downsample.0")
        s = sd[bn + ".weight"] / torch.sqrt(sd[k] + R.BN_EPS)
        folded[conv + ".weight"] = sd[conv + ".weight"] * s.view(-1, 1, 1, 1)
        bias[conv] = sd[bn + ".bias"] - sd[bn + ".running_mean"] * s

    def conv(name, t, stride=1, pad=0):
        return F.conv2d(t, folded[name + ".weight"], bias[name], stride=stride, padding=pad)

    t = F.max_pool2d(F.relu(conv("layer_base.0", x, 2, 3)), 3, 2, 1)
    for i in range(4):
        for j in range(2):
            pre = "layer_stages.%d.%d" % (i, j)
            stride = 2 if (i > 0 and j == 0) else 1
            res = conv(pre + ".downsample.0", t, stride) if pre + ".downsample.0.weight" in sd else t
            t = F.relu(conv(pre + ".conv2", F.relu(conv(pre + ".conv1", t, stride, 1)), 1, 1) + res)
        assert torch.allclose(t, want[i], rtol=1e-4, atol=1e-4 * want[i].abs().max().item())


def test_oracle_head_contraction_is_an_identity():
    """upscale_side_prep (16 -> 16 transposed conv), concat and the 1x1 fuse equal ONE [k][k][16] filter per scale."""
    sd = R.make_state_dict(18, 3, seed=5)
    h, w = 70, 101
    g = torch.Generator().manual_seed(2)
    fused = sd["layer_fuse.bias"].view(1, 1, 1, 1).clone()
    ups = []
    for s, (a, b) in enumerate(((18, 26), (9, 13), (5, 7), (3, 4))):
        side = torch.randn(1, 16, a, b, generator=g)
        f = 4 << s
        wt = sd["upscale_side_prep.%d.weight" % s]
        ups.append(R.center_crop(F.conv_transpose2d(side, wt, stride=f), h, w))
        G = torch.einsum("o,iokl->ikl", sd["layer_fuse.weight"][0, 16 * s:16 * s + 16, 0, 0], wt).unsqueeze(1)  # [16,1,k,k]
        fused = fused + R.center_crop(F.conv_transpose2d(side, G, stride=f), h, w)
    want = F.conv2d(torch.cat(ups, 1), sd["layer_fuse.weight"], sd["layer_fuse.bias"])
    assert torch.allclose(fused, want, rtol=1e-4, atol=1e-4)


def test_oracle_output_geometry_and_batch_independence():
    sd = R.make_state_dict(34, 3, seed=1)
    x = 50.0 * torch.randn(2, 3, 65, 97, generator=torch.Generator().manual_seed(3))
    outs = R.forward(sd, x)
    assert len(outs) == 5 and all(o.shape == (2, 1, 65, 97) for o in outs)
    one = R.forward(sd, x[1:])
    assert all(torch.allclose(a[1:], b, rtol=1e-4, atol=1e-4 * b.abs().max().item()) for a, b in zip(outs, one))
    assert R.crop_offsets(1084, 1080) == (2, 2) and R.crop_offsets(43, 40) == (1, 2)     # floor in front, ceil behind


def test_host_side_size_queries():
    import fosvos_hip
    from fosvos_hip import Conv2dDesc, ResnetBlock, ResnetNet
    lib = fosvos_hip.lib()
    assert lib.fosvos_conv2d_packed_dwords(16, 16, 3) == 2 * 9 * 4 * 16 + 64
    assert lib.fosvos_conv2d_packed_dwords(21, 13, 3) == 2 * 9 * 4 * 24 + 64          # both sides padded to 8
    assert lib.fosvos_conv2d_packed_dwords(32, 16, 1) == 2 * 1 * 4 * 32 + 64
    assert lib.fosvos_conv2d_packed_dwords(16, 16, 5) == 0
    assert lib.fosvos_conv2d_bias_elems(21) == 64 + 64
    assert lib.fosvos_conv7x7_packed_elems(16) == 147 * 16 + 64 + 6 * 4 * 16 * 8 // 2   # fp32 image, then bf16 MFMA fragments
    assert lib.fosvos_conv7x7_packed_elems(24) == 147 * 24 + 64 + 6 * 4 * 32 * 8 // 2

    # arena arithmetic of the native loop on a ResNet-18 / exponent 2 shaped net (fake, never dereferenced pointers)
    fake = 0x1000
    planes = [16, 32, 64, 128]
    blocks = (ResnetBlock * 8)()
    inpl = 16
    for i, p in enumerate(planes):
        for j in range(2):
            b = blocks[2 * i + j]
            stride = 2 if (i > 0 and j == 0) else 1
            b.n_convs = 2
            b.conv[0] = Conv2dDesc(fake, fake, inpl, p, 3, stride)
            b.conv[1] = Conv2dDesc(fake, fake, p, p, 3, 1)
            b.has_down = int(stride != 1 or inpl != p)
            if b.has_down:
                b.down = Conv2dDesc(fake, fake, inpl, p, 1, stride)
            inpl = p
    net = ResnetNet()
    net.first_w, net.first_b, net.first_co = fake, fake, 16
    net.blocks = ctypes.cast(blocks, ctypes.POINTER(ResnetBlock))
    for s in range(4):
        net.blocks_per_stage[s] = 2
        net.side[s] = Conv2dDesc(fake, fake, planes[s], 16, 3, 1)
        net.stride[s] = 4 << s
    need = lib.fosvos_resnet_arena_bytes(ctypes.byref(net), 1, 1080, 1920)
    first = 540 * 960 * 16 * 2
    slot = 270 * 480 * 16 * 2
    sides = sum(h * w * 16 * 4 for h, w in ((270, 480), (135, 240), (68, 120), (34, 60)))
    assert first + 4 * slot + sides <= need <= first + 4 * slot + sides + 16 * 256
    # a side branch that does not fit its stage is refused with a message, not run
    net.side[2] = Conv2dDesc(fake, fake, 48, 16, 3, 1)
    assert lib.fosvos_resnet_arena_bytes(ctypes.byref(net), 1, 1080, 1920) == 0
    assert b"side_prep 2 expects 48" in lib.fosvos_last_error()
