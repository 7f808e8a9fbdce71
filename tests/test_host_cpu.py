"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header declares,
the drop-in module surface (state_dict, optimizer recipe, flags), and loud failure without a GPU."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))

from oracle import osvos_ref as O  # noqa: E402


def test_library_exports_every_header_symbol():
    import fosvos_hip
    header = open(fosvos_hip.HEADER_PATH).read()
    declared = set(re.findall(r"\b(fosvos_[a-z0-9_]+)\s*\(", header))
    declared -= {"fosvos_sgd_entry"}
    assert declared == set(fosvos_hip.SIGNATURES), (declared ^ set(fosvos_hip.SIGNATURES))
    lib = fosvos_hip.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.fosvos_abi_version() == fosvos_hip.ABI_VERSION
    assert lib.fosvos_build_arch() == b"gfx950"
    # pure host-side queries (no GPU needed)
    assert lib.fosvos_packed_weight_elems(64, 64) == 64 * 64 * 9
    assert lib.fosvos_packed_weight_elems(16, 40) == 16 * 64 * 9     # contraction side padded to 32
    assert lib.fosvos_conv3x3_wgrad_workspace_bytes(1, 480, 854, 64, 64) > 0
    assert lib.fosvos_conv3x3_workspace_bytes(1, 480, 854, 64, 64) == 0      # enough pixel tiles: no split-K
    assert lib.fosvos_conv3x3_workspace_bytes(1, 30, 54, 512, 512) > 0       # stage 5: split-K slabs
    assert lib.fosvos_head_bwd_workspace_bytes(1, 480, 854) > 0
    assert lib.fosvos_cbce_workspace_bytes(480 * 854) > 0
    assert lib.fosvos_ctx_device(None) == -1


def _plan(n, h, w, ci, co):
    import ctypes
    import fosvos_hip
    info = fosvos_hip.Conv3x3PlanInfo()
    fosvos_hip.check(fosvos_hip.lib().fosvos_conv3x3_plan(n, h, w, ci, co, ctypes.byref(info)), "conv3x3_plan")
    return (info.tile_h, info.tile_w, info.tile_co, info.k_splits, info.workgroups)


def test_conv_plan_of_the_480p_step():
    """Which igemm instantiation each layer of the 854x480 step gets (host arithmetic of fosvos_conv3x3_plan): the GPU op
    tests assert the same query for their cases, so this table is what ties them to the benchmarked configuration."""
    # (N, H, W, Ci, Co) -> tile: stages 1-3 of one frame and every stage-1-4 layer of a five-frame pass run 256-pixel tiles
    assert _plan(1, 480, 854, 64, 64)[:4] == (8, 32, 64, 1)
    assert _plan(1, 240, 427, 64, 128)[:4] == (16, 16, 64, 1)
    assert _plan(1, 240, 427, 128, 128)[:4] == (16, 16, 64, 1)
    assert _plan(5, 120, 214, 128, 256)[:4] == (8, 32, 64, 1)
    assert _plan(5, 60, 107, 512, 512)[:4] == (16, 16, 64, 1)
    assert _plan(5, 30, 54, 512, 512)[:4] == (8, 32, 64, 1)        # stage 5 of five frames: 320 workgroups of 256 pixels
    assert _plan(3, 30, 54, 512, 512)[:4] == (8, 16, 64, 1)        # ... of a three-frame forward chain: 384 128-pixel tiles
    assert _plan(1, 120, 214, 64, 64)[:3] == (8, 16, 64)           # 105 blocks of 256 px: the 128-pixel tile
    assert _plan(1, 30, 54, 512, 512)[3] > 1                       # stage 5 of one frame: split-K
    assert _plan(1, 240, 427, 128, 16)[:3] == (8, 32, 16)          # side_prep at large maps
    import ctypes
    import fosvos_hip
    tiles, wgs = ctypes.c_int(), ctypes.c_int()
    fosvos_hip.check(fosvos_hip.lib().fosvos_conv3x3_first_plan(1, 480, 854, ctypes.byref(tiles), ctypes.byref(wgs)), "plan")
    assert (tiles.value, wgs.value) == (60 * 27, 1024)             # the persistent conv1_1 loop iterates at 480x854


def test_module_surface_matches_reference_contract():
    from networks.osvos_vgg import OSVOS_VGG
    from fosvos_hip import engine
    net = OSVOS_VGG(pretrained=0)
    spec = O.state_dict_spec()
    assert list(net.state_dict().keys()) == list(spec.keys()) == engine.PARAM_NAMES
    for k, v in net.state_dict().items():
        assert tuple(v.shape) == spec[k], k
    # the reference's initialisation: N(0, 1e-3) convs, zero biases, diagonal bilinear deconvs
    assert abs(net.stages[2][1].weight.std().item() - 1e-3) < 5e-5
    assert net.stages[2][1].bias.abs().max().item() == 0
    for i in range(4):
        assert torch.equal(net.upscale[i].weight.data, O.bilinear_deconv_weight(16, 4 << i))
        assert torch.equal(net.upscale_[i].weight.data, O.bilinear_deconv_weight(1, 4 << i))
    # module indices inside the stages (pool first in stages 1..4)
    assert isinstance(net.stages[0][0], torch.nn.Conv2d) and isinstance(net.stages[1][0], torch.nn.MaxPool2d)
    assert net.stages[1][0].ceil_mode
    # whole-module pickles keep working and carry no device caches
    import pickle
    clone = pickle.loads(pickle.dumps(net))
    assert list(clone.state_dict().keys()) == list(spec.keys())
    # ... also after a training loop attached its flat gradient buffer (and possibly pending collectives) to the module
    import parallel
    named = list(net.named_parameters())
    flat = parallel.FlatGrads.attach(net, [p for _, p in named], names=[n for n, _ in named])
    flat._works.append(object())
    assert net._fosvos_flat_grads is flat
    clone = pickle.loads(pickle.dumps(net))
    assert not hasattr(clone, "_fosvos_flat_grads") and all(p.grad is None for p in clone.parameters())
    assert list(clone.state_dict().keys()) == list(spec.keys())


def test_no_cpu_fallback():
    from networks.osvos_vgg import OSVOS_VGG
    from layers.osvos_layers import class_balanced_cross_entropy_loss
    net = OSVOS_VGG(pretrained=0)
    with pytest.raises(RuntimeError, match="GPU"):
        net(torch.zeros(1, 3, 16, 16))
    with pytest.raises(RuntimeError, match="GPU"):
        class_balanced_cross_entropy_loss(torch.zeros(1, 1, 4, 4), torch.zeros(1, 1, 4, 4))


def test_optimizer_recipe_matches_reference(golden):
    from networks.osvos_vgg import OSVOS_VGG
    from util.network_provider import VGGOfflineProvider, VGGOnlineProvider
    k = golden("loops.npz")
    for mode, cls in (("online", VGGOnlineProvider), ("offline", VGGOfflineProvider)):
        prov = cls.__new__(cls)
        prov.network = OSVOS_VGG(pretrained=0)
        opt = prov.get_optimizer()
        assert isinstance(opt, torch.optim.SGD)
        names = {id(p): n for n, p in prov.network.named_parameters()}
        rows = [f"{gi}|{names[id(p)]}|{grp['lr']!r}|{grp['weight_decay']!r}|{grp['momentum']!r}"
                for gi, grp in enumerate(opt.param_groups) for p in grp["params"]]
        assert rows == [str(s) for s in k[f"groups_{mode}"]]


def test_layer_helpers_match_reference(golden):
    from layers import osvos_layers as L
    import numpy as np
    k = golden("kat.npz")
    for size in (3, 4, 5, 8, 16, 32):
        np.testing.assert_array_equal(L.upsample_filt(size), k[f"filt_{size}"])
    src = torch.from_numpy(k["crop_src"])
    for h, w in k["crop_cases"]:
        np.testing.assert_array_equal(L.center_crop(src, int(h), int(w)).numpy(), k[f"crop_{h}_{w}"])
    for c, size in ((16, 4), (1, 8), (3, 16)):
        lay = torch.nn.ConvTranspose2d(c, c, size, stride=size // 2, bias=False)
        np.testing.assert_array_equal(L.interp_surgery(lay).numpy(), k[f"surgery_{c}_{size}"])


def test_cli_flags():
    from util import args_helper
    a = args_helper.parse_args(True, ["--gpu-id", "0", "-s", "blackswan", "-sg", "1", "-sgs", "4", "--variant-online", "2",
                                      "--no-testing", "--eval-speeds"])
    assert (a.gpu_id, a.sequence_name, a.sequence_group, a.sequence_group_size, a.variant_online) == (0, "blackswan", 1, 4, 2)
    assert a.is_training and not a.is_testing and a.eval_speeds and a.network == "vgg16"
    b = args_helper.parse_args(False, [])
    assert not hasattr(b, "sequence_name") and b.variant_offline is None


def test_sequence_sharding_matches_reference_rule():
    import parallel
    import train_online
    seqs = train_online.sequences_val
    assert len(seqs) == 20
    parts = [parallel.shard_sequences(seqs, g, 8) for g in range(8)]
    assert sorted(sum(parts, [])) == sorted(seqs)
    assert parts[3] == [s for i, s in enumerate(seqs) if i % 8 == 3]  # src/train_online.py:184-186
    assert parallel.shard_sequences(seqs, None, None) == seqs
    assert parallel.split_accumulation(8, 8) == 1 and parallel.split_accumulation(10, 2) == 5
    with pytest.raises(ValueError):
        parallel.split_accumulation(5, 8)


def test_png_bytescale_known_answers():
    """scipy.misc.imsave's scaling (what src/util/experiment_helper.py:64 relied on): stretch to the map's own range,
    round half up; a constant map becomes zeros."""
    from util.experiment_helper import bytescale
    assert bytescale(np.array([[0.2, 0.7], [0.45, 0.2]])).tolist() == [[0, 255], [128, 0]]
    assert bytescale(np.full((2, 3), 0.37)).tolist() == [[0, 0, 0], [0, 0, 0]]
    assert bytescale(np.array([[0.0, 1.0, 0.5, 0.25]])).tolist() == [[0, 255, 128, 64]]
