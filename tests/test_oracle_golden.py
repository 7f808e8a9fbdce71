"""The CPU oracle (oracle/osvos_ref.py) against fixtures produced by the reference itself
(oracle/make_golden.py).  Runs on CPU; this is what pins parity."""
import numpy as np
import pytest
import torch

from oracle import osvos_ref as O

RTOL = 1e-5  # fp32 restatement vs the reference on the same torch build


def T(a):
    return torch.from_numpy(np.asarray(a))


def check_digest(t, m, idx, smp, rtol=RTOL, scale=None):
    f = t.detach().reshape(-1).double()
    got = np.array([f.sum().item(), f.abs().sum().item(), (f * f).sum().item()])
    slack = 0.0 if scale is None else scale * f.numel()  # per-element absolute slack, summed
    ref_scale = max(abs(m[1]), 1e-30)
    assert abs(got[0] - m[0]) <= rtol * ref_scale * 10 + slack
    assert abs(got[1] - m[1]) <= rtol * ref_scale * 10 + slack
    assert abs(got[2] - m[2]) <= rtol * max(m[2], 1e-30) * 10 + slack * np.sqrt(max(m[2], 1e-30))
    s = t.detach().reshape(-1)[T(idx)].double().numpy()
    tol = rtol * (np.abs(smp).max() + 1e-30) * 10 if scale is None else scale
    np.testing.assert_allclose(s, smp, rtol=rtol * 10, atol=tol)


def test_bilinear_kernel(golden):
    k = golden("kat.npz")
    for size in (3, 4, 5, 8, 16, 32):
        np.testing.assert_array_equal(O.bilinear_kernel(size), k[f"filt_{size}"])
    assert list(O.bilinear_kernel(4)[0]) == [.0625, .1875, .1875, .0625]
    for size, total in ((4, 4), (8, 16), (16, 64), (32, 256)):
        assert O.bilinear_kernel(size).sum() == total
    for c, size in ((16, 4), (1, 8), (3, 16)):
        np.testing.assert_array_equal(O.bilinear_deconv_weight(c, size).numpy(), k[f"surgery_{c}_{size}"])


def test_center_crop(golden):
    k = golden("kat.npz")
    src = T(k["crop_src"])
    for h, w in k["crop_cases"]:
        np.testing.assert_array_equal(O.center_crop(src, int(h), int(w)).numpy(), k[f"crop_{h}_{w}"])
    # the odd pixel comes off the bottom/right
    assert O.crop_offsets(11, 8) == (1, 2)
    assert O.crop_offsets(482, 480) == (1, 1)
    assert O.crop_offsets(880, 854) == (13, 13)
    assert O.crop_offsets(860, 854) == (3, 3)
    assert O.crop_offsets(6, 1) == (2, 3)


@pytest.mark.parametrize("tag", ["kat12", "allneg", "allpos", "extreme", "rand", "soft"])
def test_cbce(golden, tag):
    k = golden("kat.npz")
    x, y = T(k[f"loss_{tag}_x"]), T(k[f"loss_{tag}_y"])
    ls = O.cbce_loss(x, y, size_average=False)
    la = O.cbce_loss(x, y, size_average=True)
    np.testing.assert_allclose(ls.item(), k[f"loss_{tag}_sum"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(la.item(), k[f"loss_{tag}_avg"], rtol=2e-6, atol=1e-7)
    g = O.cbce_loss_grad(x, y, size_average=False)
    np.testing.assert_allclose(g.numpy(), k[f"loss_{tag}_grad"], rtol=1e-5, atol=1e-7)
    xr = x.clone().requires_grad_(True)
    (ga,) = torch.autograd.grad(O.cbce_loss(xr, y, size_average=False), xr)
    np.testing.assert_allclose(ga.numpy(), k[f"loss_{tag}_grad"], rtol=1e-5, atol=1e-7)


def test_cbce_known_values():
    x = torch.linspace(-3, 3, 12).view(1, 1, 3, 4)
    y = torch.tensor([0, 0, 1, 0, 0, 1, 1, 0, 0, 0, .5, .49]).view(1, 1, 3, 4)
    assert abs(O.cbce_loss(x, y, False).item() - 5.221406936645508) < 1e-5
    assert abs(O.cbce_loss(x, y, True).item() - 0.435117244720459) < 1e-6
    assert abs(O.cbce_loss_grad(x, y, False).reshape(-1)[2].item() - (-0.58061135)) < 1e-6
    assert O.cbce_loss(x, torch.zeros_like(x), False).item() == 0.0
    xe = torch.tensor([-100., 100., 0.]).view(1, 1, 1, 3)
    ye = torch.tensor([1., 0., 1.]).view(1, 1, 1, 3)
    assert abs(O.cbce_loss(xe, ye, False).item() - 100.2310562) < 1e-4


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e"])
def test_layer_stacks(golden, tag):
    """pool(ceil) + conv3x3 + bias + relu stacks built by the reference's _make_layers_osvos."""
    import torch.nn.functional as F
    k = golden("stacks.npz")
    cfg = [int(v) for v in k[f"{tag}_cfg"]]
    x = T(k[f"{tag}_x"]).clone().requires_grad_(True)
    params = []
    h = x
    pi = 0
    for v in cfg:
        if v < 0:
            h = F.max_pool2d(h, 2, 2, ceil_mode=True)
        else:
            w = T(k[f"{tag}_p{pi}"]).clone().requires_grad_(True)
            b = T(k[f"{tag}_p{pi + 1}"]).clone().requires_grad_(True)
            pi += 2
            params += [w, b]
            h = F.relu(F.conv2d(h, w, b, padding=1))
    np.testing.assert_allclose(h.detach().numpy(), k[f"{tag}_y"], rtol=1e-5, atol=1e-5)
    grads = torch.autograd.grad(h, [x] + params, T(k[f"{tag}_gy"]))
    np.testing.assert_allclose(grads[0].numpy(), k[f"{tag}_gx"], rtol=1e-4, atol=1e-4)
    for i, g in enumerate(grads[1:]):
        np.testing.assert_allclose(g.numpy(), k[f"{tag}_gp{i}"], rtol=1e-4, atol=1e-3)


def test_state_dict_spec(golden):
    k = golden("net.npz")
    spec = O.state_dict_spec()
    assert list(spec.keys()) == [str(s) for s in k["keys"]]
    assert [str(tuple(v)) for v in spec.values()] == [str(s) for s in k["shapes"]]
    assert len(spec) == 52
    assert sum(int(np.prod(s)) for s in spec.values()) == 15267157
    sd = O.make_state_dict(0, "reference")
    for i in range(4):
        np.testing.assert_array_equal(sd[f"upscale.{i}.weight"][3, 3].numpy(), k[f"init_upscale_{i}_diag"])
        assert sd[f"upscale.{i}.weight"][3, 5].abs().max().item() == k[f"init_upscale_{i}_offdiag_absmax"] == 0
        np.testing.assert_array_equal(sd[f"upscale_.{i}.weight"][0, 0].numpy(), k[f"init_upscale__{i}"])
    assert abs(sd["stages.2.1.weight"].std().item() - float(k["init_conv_std"])) < 5e-5
    assert float(k["init_bias_absmax"]) == 0.0 == sd["stages.2.1.bias"].abs().max().item()


@pytest.mark.parametrize("tag", ["s", "r"])
def test_full_network(golden, tag):
    k = golden("net.npz")
    n, h, w = (int(v) for v in k[f"{tag}_shape"])
    sd = O.make_state_dict(int(k[f"{tag}_seed"]))
    x, gt = O.synthetic_frame(n, h, w, seed=int(k[f"{tag}_frame_seed"]))
    params = O.leaf_params(sd)
    outs = O.forward(params, x)
    assert len(outs) == 5
    losses = []
    for i, o in enumerate(outs):
        ref = k[f"{tag}_out{i}"]
        assert tuple(o.shape) == ref.shape == (n, 1, h, w)
        np.testing.assert_allclose(o.detach().numpy(), ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
        li = O.cbce_loss(o, gt, size_average=False)
        losses.append(li)
        np.testing.assert_allclose(li.item(), float(k[f"{tag}_loss{i}"]), rtol=1e-5)
    # online objective
    names = list(params.keys())
    g_on = torch.autograd.grad(losses[-1], list(params.values()), retain_graph=True, allow_unused=True)
    nograd = set(str(s) for s in k[f"{tag}_on_nograd"])
    assert nograd == {nm for nm, g in zip(names, g_on) if g is None}
    assert nograd == {f"score_dsn.{i}.{p}" for i in range(4) for p in ("weight", "bias")} | {
        f"upscale_.{i}.weight" for i in range(4)}
    for nm, g in zip(names, g_on):
        if g is not None:
            check_digest(g, k[f"{tag}_on_g_{nm}_m"], k[f"{tag}_on_g_{nm}_i"], k[f"{tag}_on_g_{nm}_s"], rtol=1e-4)
    g_off = torch.autograd.grad((1 - 60 / 240) * sum(losses[:-1]) + losses[-1], list(params.values()))
    for nm, g in zip(names, g_off):
        check_digest(g, k[f"{tag}_off_g_{nm}_m"], k[f"{tag}_off_g_{nm}_i"], k[f"{tag}_off_g_{nm}_s"], rtol=1e-4)


def test_optimizer_groups(golden):
    k = golden("loops.npz")
    for mode in ("online", "offline"):
        params = O.leaf_params(O.make_state_dict(5))
        opt = O.make_sgd(params, mode)
        names = {id(p): n for n, p in params.items()}
        rows = []
        for gi, grp in enumerate(opt.param_groups):
            for p in grp["params"]:
                rows.append(f"{gi}|{names[id(p)]}|{grp['lr']!r}|{grp['weight_decay']!r}|{grp['momentum']!r}")
        assert rows == [str(s) for s in k[f"groups_{mode}"]]


@pytest.mark.parametrize("tag,lr", [("lr1e-8", 1e-8), ("lr1e-9", 1e-9)])
def test_online_loop(golden, tag, lr):
    k = golden("loops.npz")
    sd = O.make_state_dict(6)
    f0 = O.synthetic_frame(1, 48, 86, seed=21)
    f1 = O.synthetic_frame(1, 40, 70, seed=22)
    losses, final = O.online_loop(sd, [f0[0], f1[0]], [f0[1], f1[1]], 10, 5, lr=lr)
    np.testing.assert_allclose(losses, k[f"online_{tag}_loss"], rtol=2e-5)
    for nm in sd:
        d = final[nm].double() - sd[nm].double()
        m = k[f"online_{tag}_delta_{nm}_m"]
        # the applied update sits near the fp32 resolution of the weight: allow 2 ulps of it on top
        ulp = float(np.spacing(np.float32(sd[nm].abs().max().item())))
        check_digest(d, m, k[f"online_{tag}_delta_{nm}_i"], k[f"online_{tag}_delta_{nm}_s"], rtol=2e-3,
                     scale=2e-3 * (np.abs(k[f"online_{tag}_delta_{nm}_s"]).max() + 1e-30) + 2 * ulp)
    np.testing.assert_allclose(final["fuse.weight"].numpy(), k[f"online_{tag}_fuse_weight"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(final["stages.0.0.bias"].numpy(), k[f"online_{tag}_stage00_bias"], rtol=1e-6, atol=1e-9)
    # frozen / unoptimised tensors do not move (src/util/network_provider.py:154-155, score_dsn absent)
    for nm in sd:
        if nm.startswith(("upscale", "score_dsn")):
            assert torch.equal(final[nm], sd[nm])


def test_finetune_trajectory(golden):
    """The oracle's online loop follows the REFERENCE's 60-iteration fine-tune (oracle/make_golden.py section 6): losses,
    the held-out logits and mask after training, and the full applied delta of every small tensor."""
    k = golden("trajectory.npz")
    for key, val in O.TRAJ.items():
        assert float(k[key]) == float(val), key
    sd, frames, (xh, gh) = O.trajectory_inputs()
    losses, final = O.online_loop(sd, [f[0] for f in frames], [f[1] for f in frames], O.TRAJ["iters"], O.TRAJ["avg"],
                                  lr=O.TRAJ["lr"])
    np.testing.assert_allclose(losses, k["loss"], rtol=1e-3)
    assert k["loss"][-1] < 0.1 * k["loss"][0]  # the schedule does train
    with torch.no_grad():
        held = O.forward(final, xh)[-1][0, 0]
    ref = torch.from_numpy(k["heldout_logits"])
    assert (held - ref).abs().max().item() <= 1e-3 * ref.abs().max().item()
    ref_mask = torch.from_numpy(np.unpackbits(k["heldout_mask_bits"])[: ref.numel()].reshape(ref.shape)).bool()
    assert torch.equal(ref_mask, ref >= 0)
    gt_mask = torch.from_numpy(np.unpackbits(k["heldout_gt_bits"])[: ref.numel()].reshape(ref.shape)).bool()
    assert torch.equal(gt_mask, gh[0, 0] > 0.5)
    assert abs(O.mask_iou(held >= 0, ref_mask) - 1.0) <= 1e-3
    start = torch.from_numpy(k["heldout_logits_start"])
    assert O.mask_iou(start >= 0, gt_mask) < 0.05 < 0.8 < O.mask_iou(ref_mask, gt_mask)  # the mask moved onto the object
    for nm in k["full_tensors"]:
        nm = str(nm)
        d = (final[nm].double() - sd[nm].double()).numpy()
        ref_d = k[f"delta_{nm}"]
        scale = np.abs(ref_d).max()
        if nm.startswith(("upscale", "score_dsn")):
            assert scale == 0 and np.abs(d).max() == 0
            continue
        ulp = float(np.spacing(np.float32(sd[nm].abs().max().item())))
        # 60 iterations of fp32 arithmetic in two statements of the same graph (nn modules vs functional calls): the
        # summation orders of the threaded CPU kernels differ, and twelve optimizer steps carry that forward
        assert np.abs(d - ref_d).max() <= 1e-2 * scale + 2 * ulp, nm
        assert np.linalg.norm(d - ref_d) <= 5e-3 * np.linalg.norm(ref_d) + 2 * ulp * np.sqrt(d.size), nm
    for nm in k["digest_tensors"]:
        nm = str(nm)
        d = final[nm].double() - sd[nm].double()
        ulp = float(np.spacing(np.float32(sd[nm].abs().max().item())))
        smp = k[f"delta_{nm}_s"]
        got = d.reshape(-1)[torch.from_numpy(k[f"delta_{nm}_i"])].numpy()
        assert np.abs(got - smp).max() <= 1e-2 * np.abs(smp).max() + 2 * ulp, nm
        assert np.linalg.norm(got - smp) <= 5e-3 * np.linalg.norm(smp) + 2 * ulp * np.sqrt(got.size), nm


def test_offline_loop(golden):
    k = golden("loops.npz")
    sd = O.make_state_dict(8)
    x, gt = O.synthetic_frame(2, 33, 47, seed=23)
    trace, final = O.offline_loop(sd, [x], [gt], 4, epoch=60, n_epochs=240, avg_grad_every_n=2, lr=1e-6)
    np.testing.assert_allclose(np.array(trace), k["offline_loss"], rtol=2e-5)
    for nm in sd:
        d = final[nm].double() - sd[nm].double()
        ulp = float(np.spacing(np.float32(sd[nm].abs().max().item())))
        check_digest(d, k[f"offline_delta_{nm}_m"], k[f"offline_delta_{nm}_i"], k[f"offline_delta_{nm}_s"],
                     rtol=2e-3, scale=2e-3 * (np.abs(k[f"offline_delta_{nm}_s"]).max() + 1e-30) + 2 * ulp)


def test_e2e_frame(golden):
    """One 854x480 frame through the oracle vs the reference's stored logits/mask."""
    k = golden("e2e_480x854.npz")
    sd = O.make_state_dict(int(k["seed"]))
    x, gt = O.synthetic_frame(1, 480, 854, seed=int(k["frame_seed"]))
    with torch.no_grad():
        outs = O.forward(sd, x)
    fused = outs[-1][0, 0]
    ref = T(k["logits_f16"]).float()
    amax = float(k["logits_absmax"])
    assert (fused - ref).abs().max().item() <= 2e-3 * amax  # fp16 storage resolution
    ref_mask = T(np.unpackbits(k["mask_bits"])[: 480 * 854].reshape(480, 854)).bool()
    mask = O.logits_to_mask(fused)
    band = fused.abs() > 1e-4 * amax
    assert torch.equal(mask[band], ref_mask[band])
    assert abs(O.mask_iou(mask, ref_mask) - 1.0) <= 1e-3
    np.testing.assert_allclose(O.cbce_loss(outs[-1], gt, False).item(), float(k["loss_fused_sum"]), rtol=1e-5)
    for i in range(4):
        check_digest(outs[i], k[f"side{i}_m"], k[f"side{i}_i"], k[f"side{i}_s"], rtol=1e-4)


def test_backward_480x854(golden):
    """The oracle's backward pass at the BASELINE frame size against the reference's (oracle/make_golden.py section 7): the
    fused loss and every gradient - small tensors element by element, the larger ones on their 4096 strided samples and
    float64 moments."""
    k = golden("bwd_480x854.npz")
    sd = O.make_state_dict(int(k["seed"]))
    x, gt = O.synthetic_frame(1, 480, 854, seed=int(k["frame_seed"]))
    params = O.leaf_params(sd)
    loss = O.cbce_loss(O.forward(params, x)[-1], gt, size_average=False)
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(k["loss_fused_sum"]), rtol=1e-5)
    for nm in k["nograd"]:
        assert params[str(nm)].grad is None, nm
    for nm in list(k["full_tensors"]) + list(k["dense_tensors"]):
        nm = str(nm)
        g = params[nm].grad.double().reshape(-1)
        if f"g_{nm}" in k.files:
            ref = torch.from_numpy(k[f"g_{nm}"]).double().reshape(-1)
            got = g
        else:
            ref = torch.from_numpy(k[f"g_{nm}_s"]).double()
            got = g[torch.from_numpy(k[f"g_{nm}_i"])]
        # two statements of one fp32 graph over 410 k pixels (nn modules vs functional calls, threaded reductions)
        assert float((got - ref).norm()) <= 2e-3 * float(ref.norm()) + 1e-30, nm
        m = k[f"g_{nm}_m"]
        np.testing.assert_allclose(float(g.abs().sum()), m[1], rtol=2e-3)
        np.testing.assert_allclose(float((g * g).sum()), m[2], rtol=4e-3)


def test_finetune_trajectory_480x854_first_cycle(golden):
    """The 480x854 fine-tune fixture (oracle/make_golden.py section 8; 30 reference iterations): the oracle follows its first
    accumulation cycle, optimizer step and the next iteration (6 forward / backward passes: about a minute of CPU time;
    the GPU test runs the whole schedule), and the fixture is self-consistent (schedule constants, mask bits = sign of the
    stored logits outside fp16 resolution, the run trains)."""
    k = golden("trajectory_480x854.npz")
    for key, val in O.TRAJ_FULL.items():
        assert float(k[key]) == float(val), key
    assert len(k["loss"]) == O.TRAJ_FULL["iters"] and k["loss"][-1] < 0.15 * k["loss"][0]
    sd, frames, (xh, gh) = O.trajectory_inputs(O.TRAJ_FULL)
    losses, _ = O.online_loop(sd, [f[0] for f in frames], [f[1] for f in frames], O.TRAJ_FULL["avg"] + 1,
                              O.TRAJ_FULL["avg"], lr=O.TRAJ_FULL["lr"])
    np.testing.assert_allclose(losses, k["loss"][:len(losses)], rtol=1e-3)
    ref = torch.from_numpy(k["heldout_logits_f16"].astype(np.float32))
    n = ref.numel()
    ref_mask = torch.from_numpy(np.unpackbits(k["heldout_mask_bits"])[:n].reshape(ref.shape)).bool()
    sure = ref.abs() > 2e-3 * float(k["heldout_logits_absmax"])
    assert torch.equal((ref >= 0)[sure], ref_mask[sure])
    gt_mask = torch.from_numpy(np.unpackbits(k["heldout_gt_bits"])[:n].reshape(ref.shape)).bool()
    assert torch.equal(gt_mask, gh[0, 0] > 0.5)
    start_mask = torch.from_numpy(np.unpackbits(k["heldout_start_mask_bits"])[:n].reshape(ref.shape)).bool()
    assert O.mask_iou(start_mask, gt_mask) < 0.05 < 0.9 < O.mask_iou(ref_mask, gt_mask)  # the mask moved onto the object
    assert len(k["full_tensors"]) + len(k["digest_tensors"]) == 52


def test_bf16_emulation_mode_is_close_to_fp32():
    """The bf16-emulating oracle mode (used to separate precision scheme from kernel correctness in the GPU
    tests) is the same graph plus rounding: its logits stay within 2 % of the fp32 logit range."""
    sd = O.make_state_dict(3)
    x, gt = O.synthetic_frame(1, 48, 86, seed=103)
    ref = O.forward(sd, x)
    emu = O.forward(sd, x, emulate_bf16=True)
    for a, b in zip(emu, ref):
        assert (a - b).abs().max().item() <= 2e-2 * b.abs().max().item()
    params = O.leaf_params(sd)
    O.cbce_loss(O.forward(params, x, emulate_bf16=True)[-1], gt, False).backward()
    g = params["stages.2.3.weight"].grad
    assert g is not None and torch.isfinite(g).all() and g.abs().sum() > 0
