"""world_size-2 gloo tests (CPU) that drive the SHIPPED loop functions - ``train_online._train`` and
``train_offline._train`` - in data-parallel mode and compare the weights they leave with the same function run in one
process on the whole work.  The HIP network and loss need a GPU, so both are replaced by CPU stand-ins with the same
surface: a small torch module with OSVOS_VGG's attribute names (so the gradient buckets of ``parallel.VGG_BUCKETS`` apply
as they do to the real net) and the oracle's class-balanced loss.  What is under test is the wiring: FlatGrads views and
buckets, the accumulation split (online) / batch split with global class counts (offline), the bucketed asynchronous
all-reduce and where it is joined, the optimizer step on the reduced buffer."""
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class TinyOSVOS(nn.Module):
    """CPU stand-in with OSVOS_VGG's module names and the loop-local switches `_train` flips."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.upscale = nn.ModuleList([nn.ConvTranspose2d(2, 2, 2, 2, bias=False) for _ in range(4)])  # frozen by recipe
        self.stages = nn.ModuleList([nn.Sequential(nn.Conv2d(3 if i == 0 else 4, 4, 3, padding=1), nn.ReLU())
                                     for i in range(5)])
        self.side_prep = nn.ModuleList([nn.Conv2d(4, 2, 3, padding=1) for _ in range(4)])
        self.score_dsn = nn.ModuleList([nn.Conv2d(2, 1, 1) for _ in range(4)])
        self.fuse = nn.Conv2d(8, 1, 1)
        self.accumulate_grads_in_place = False
        self.compute_side_outputs = True
        self.defer_wgrad_join = False

    def join_gradients(self):
        pass

    def forward(self, x):
        sides, outs = [], []
        for i, st in enumerate(self.stages):
            x = st(x)
            if i > 0:
                s = self.side_prep[i - 1](x)
                sides.append(s)
                outs.append(self.score_dsn[i - 1](s))
        outs.append(self.fuse(torch.cat(sides, dim=1)))
        return outs


def _frames(n, batch=1):
    g = torch.Generator().manual_seed(5)
    return [{"image": torch.randn(batch, 3, 10, 12, generator=g),
             "gt": (torch.rand(batch, 1, 10, 12, generator=g) > 0.7).float()} for _ in range(n)]


def _cbce(output, label, size_average=True, batch_counts=None):
    """The oracle's loss; with batch_counts the class weights come from the whole data-parallel batch."""
    from oracle import osvos_ref as O
    if batch_counts is None:
        return O.cbce_loss(output, label, size_average=size_average)
    y = (label >= 0.5).float()
    n_pos, n_tot = batch_counts[0].float(), batch_counts[1].float()
    x = output
    val = torch.clamp(x, min=0) - x * y + torch.log1p(torch.exp(-x.abs()))
    loss = (n_tot - n_pos) / n_tot * (y * val).sum() + n_pos / n_tot * ((1 - y) * val).sum()
    return loss / n_tot if size_average else loss


class _Prov:
    name = "tiny"

    def __init__(self, net):
        self.network = net

    def save_model(self, *a, **k):
        pass


class _Writer:
    def add_scalar(self, *a, **k):
        pass

    def close(self):
        pass


def _sgd(net):
    return torch.optim.SGD([p for n, p in net.named_parameters() if not n.startswith("upscale")], lr=1e-2, momentum=0.9)


def _run_online(loader, avg, dp):
    import train_online
    train_online.class_balanced_cross_entropy_loss = _cbce
    train_online.data_parallel = dp
    net = TinyOSVOS()
    train_online._train(_Prov(net), loader, _sgd(net), _Writer(), "tiny", 0, 1, avg, 10 ** 9)
    return net


def _run_offline(loader, avg, dp):
    import train_offline
    train_offline.class_balanced_cross_entropy_loss = _cbce
    train_offline.data_parallel = dp
    net = TinyOSVOS()
    train_offline._train(_Prov(net), loader, None, _sgd(net), _Writer(), 0, 2, avg, 10 ** 9, False, 5)
    return net


def _worker(rank, world, port, mode, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import parallel
    assert parallel.init_distributed("gloo")
    if mode == "online":     # accumulation split: rank r runs micro-batches r, r + world, ... of every cycle
        net = _run_online(_frames(8)[rank::world], 4, True)
    else:                    # batch split: rank r holds sample r of every batch of `world` samples
        full = _frames(4, batch=world)
        net = _run_offline([{k: v[rank:rank + 1] for k, v in b.items()} for b in full], 2, True)
    flat_ok = all(p.grad is None or p.grad.is_contiguous() for p in net.parameters())
    torch.save({"flat_ok": flat_ok, "sd": net.state_dict()}, os.path.join(out_dir, f"{mode}{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _launch(mode, tmp_path, world=2):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), str(world), str(port), mode, str(tmp_path)])
             for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=300) == 0
    finally:  # a failed rank must not leave its peer waiting in a collective
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    return [torch.load(os.path.join(str(tmp_path), f"{mode}{r}.pt")) for r in range(world)]


def _compare(results, ref_net):
    ref = ref_net.state_dict()
    moved = 0
    init = TinyOSVOS().state_dict()
    for r, res in enumerate(results):
        assert res["flat_ok"]
        for k, v in res["sd"].items():
            assert torch.allclose(v, ref[k], rtol=1e-5, atol=1e-7), f"rank {r}: {k} differs from the single-process run"
    for k in ref:
        assert torch.equal(results[0]["sd"][k], results[1]["sd"][k]), k  # the replicas stay bit-identical
        moved += int(not torch.equal(ref[k], init[k]))
    assert moved >= 20  # the loop did train (everything but the frozen deconvs and, online, score_dsn moved)


def test_online_train_dp_equals_single_process(tmp_path):
    """`train_online._train`, avg_grad_every_n = 4 split over 2 ranks, bucketed all-reduce: same weights as one process
    running all 8 micro-batches (src/train_online.py:92-101)."""
    _compare(_launch("online", tmp_path), _run_online(_frames(8), 4, False))


def test_offline_train_dp_equals_single_process(tmp_path):
    """`train_offline._train` with every batch of 2 split over 2 ranks: the class counts of the five losses are summed
    over the ranks, so the update equals one process running the whole batches (src/layers/osvos_layers.py:28-39 counts
    over the batch tensor; src/train_offline.py:77-110)."""
    _compare(_launch("offline", tmp_path), _run_offline(_frames(4, batch=2), 2, False))


def test_flat_grad_buckets_of_the_real_module():
    """Bucket slices of the real OSVOS_VGG parameter list: completion order, contiguous, frozen deconvs left out, every
    other trainable element covered exactly once (14,917,637 = 15,267,157 - 349,520 frozen)."""
    import parallel
    from networks.osvos_vgg import OSVOS_VGG
    net = OSVOS_VGG(pretrained=0)
    named = list(net.named_parameters())
    flat = parallel.FlatGrads([p for _, p in named], names=[n for n, _ in named])
    assert len(flat.slices) == 5
    sizes = [hi - lo for lo, hi in flat.slices]
    assert sizes[0] == 3 * (512 * 512 * 9 + 512)                           # stages.4
    assert sizes[1] == 512 * 256 * 9 + 512 + 2 * (512 * 512 * 9 + 512)     # stages.3
    assert sizes[2] == 256 * 128 * 9 + 256 + 2 * (256 * 256 * 9 + 256)     # stages.2
    covered = sum(sizes)
    frozen = sum(p.numel() for n, p in named if n.startswith("upscale"))
    assert frozen == 349520
    pad = flat.flat.numel() - sum(p.numel() for _, p in named)
    assert 0 <= covered - (15267157 - frozen) <= pad
    spans = sorted(flat.slices)
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))


def _mixed_frames():
    g = torch.Generator().manual_seed(9)
    shapes = {"A": (10, 12), "B": (8, 10), "C": (6, 14)}
    def frame(tag):
        h, w = shapes[tag]
        return {"image": torch.randn(1, 3, h, w, generator=g), "gt": (torch.rand(1, 1, h, w, generator=g) > 0.7).float()}
    return {tag + str(i): frame(tag) for tag, i in (("A", 0), ("B", 0), ("A", 1), ("C", 0), ("B", 1))}


def test_online_loop_buckets_a_cycle_by_frame_shape(monkeypatch):
    """The reference's augmentation draws a new scale per iteration (src/dataloaders/custom_transforms.py:63-76), so
    same-size frames are rarely consecutive.  `_train` buckets the micro-batches of an accumulation cycle by shape and
    runs one batched pass per shape: (i) a loader that yields the same frames already sorted by shape runs the identical
    passes - weights bit for bit; (ii) against the one-by-one order (group size 1) the update agrees to fp32 rounding;
    (iii) the logged losses stay in the reference's iteration order."""
    import train_online
    f = _mixed_frames()
    mixed = [f[k] for k in ("A0", "B0", "A1", "C0", "B1")]
    presorted = [f[k] for k in ("A0", "A1", "B0", "B1", "C0")]
    passes = []
    real_forward = TinyOSVOS.forward

    def spying_forward(self, x):
        passes.append(tuple(x.shape))
        return real_forward(self, x)

    monkeypatch.setattr(TinyOSVOS, "forward", spying_forward)

    def run(loader, group):
        monkeypatch.setenv("FOSVOS_MICROBATCH_GROUP", str(group))
        train_online.class_balanced_cross_entropy_loss = _cbce
        train_online.data_parallel = False
        net = TinyOSVOS()
        passes.clear()
        ret = train_online._train(_Prov(net), loader, _sgd(net), _Writer(), "tiny", 0, 2, 5, 10 ** 9)
        return net.state_dict(), ret["loss"], list(passes)

    w_mixed, loss_mixed, p_mixed = run(mixed, 5)
    w_sorted, loss_sorted, p_sorted = run(presorted, 5)
    w_single, loss_single, p_single = run(mixed, 1)
    assert p_mixed == p_sorted == [(2, 3, 10, 12), (2, 3, 8, 10), (1, 3, 6, 14)] * 2
    assert len(p_single) == 10 and all(s[0] == 1 for s in p_single)
    for k in w_mixed:
        assert torch.equal(w_mixed[k], w_sorted[k]), k
        assert torch.allclose(w_mixed[k], w_single[k], rtol=1e-5, atol=1e-7), k
    # every iteration of these 2 epochs is a logging point (src/train_online.py:84-90): the trace of the bucketed run equals
    # the one-by-one trace ENTRY BY ENTRY - the passes ran bucket by bucket, the log follows the reference's iteration order
    assert len(loss_mixed) == len(loss_single) == 10
    assert torch.allclose(torch.tensor(loss_mixed), torch.tensor(loss_single), rtol=1e-5)
    # frames A0 B0 A1 C0 B1 -> presorted order A0 A1 B0 B1 C0
    perm = [0, 2, 1, 4, 3]
    assert torch.allclose(torch.tensor(loss_sorted[:5]), torch.tensor([loss_mixed[i] for i in perm]), rtol=1e-5)
    assert loss_mixed[0] != loss_mixed[1]


def test_flat_grad_bucket_ids_with_a_frozen_stage():
    """A slice of the flat buffer keeps the NATIVE bucket id of its gradients (what fosvos_vgg_grad_bucket_wait takes): with
    stage 4 frozen, slice 0 is bucket 1, and a tensor outside every bucket waits for the pass's last bucket."""
    import parallel
    from networks.osvos_vgg import OSVOS_VGG
    net = OSVOS_VGG(pretrained=0)
    for p in net.stages[4].parameters():
        p.requires_grad_(False)
    named = list(net.named_parameters())
    flat = parallel.FlatGrads([p for _, p in named], names=[n for n, _ in named])
    assert flat.bucket_ids == [1, 2, 3, 4] and len(flat.slices) == 4
    seen = []
    flat2 = parallel.FlatGrads([p for _, p in named], names=[n for n, _ in named],
                               buckets=(("stages.3.",), ("stages.2.",)))  # everything else: trailing slices
    assert flat2.bucket_ids[:2] == [0, 1] and set(flat2.bucket_ids[2:]) == {1} and len(flat2.slices) >= 3
    full = parallel.FlatGrads([p for _, p in list(OSVOS_VGG(pretrained=0).named_parameters())])
    assert full.bucket_ids == [len(parallel.VGG_BUCKETS) - 1]
    del seen


def test_offline_loop_tells_the_sampler_the_epoch():
    """Data-parallel offline training draws a NEW shuffled order every epoch, like the reference's shuffle=True loader
    (src/util/io_helper.py:62-70): `_train` calls sampler.set_epoch(epoch), without which a DistributedSampler replays the
    same permutation."""
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    import train_offline
    frames = _frames(8)
    order = []

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return len(frames)

        def __getitem__(self, i):
            order.append(i)
            return {k: v[0] for k, v in frames[i].items()}

    sampler = DistributedSampler(DS(), num_replicas=2, rank=0, shuffle=True, seed=0)
    loader = DataLoader(DS(), batch_size=1, sampler=sampler, num_workers=0)
    train_offline.class_balanced_cross_entropy_loss = _cbce
    train_offline.data_parallel = False
    net = TinyOSVOS()
    train_offline._train(_Prov(net), loader, None, _sgd(net), _Writer(), 0, 3, 2, 10 ** 9, False, 5)
    epochs = [order[i:i + 4] for i in range(0, 12, 4)]
    assert len(order) == 12 and len({tuple(e) for e in epochs}) > 1, epochs


if __name__ == "__main__":
    _worker(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5])
