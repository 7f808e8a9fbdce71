"""world_size-2 gloo test (CPU) of the data-parallel fine-tune logic: the accumulation micro-batches are
spread over the ranks, the flat gradient buffer is SUM-all-reduced once per optimizer step, and the resulting
update equals the single-process loop of src/train_online.py:92-101."""
import os
import socket
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fosvos_amd"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 1, 1))


def _batches(n):
    g = torch.Generator().manual_seed(1)
    return [(torch.randn(1, 3, 12, 14, generator=g), (torch.rand(1, 1, 12, 14, generator=g) > 0.7).float()) for _ in range(n)]


def _loss(out, gt):  # any differentiable per-micro-batch loss will do for the collective logic
    return torch.nn.functional.binary_cross_entropy_with_logits(out, gt, reduction="sum")


def _reference(avg, steps):
    net = _model()
    opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9)
    data = _batches(avg * steps)
    for i, (x, y) in enumerate(data):
        (_loss(net(x), y) / avg).backward()
        if (i + 1) % avg == 0:
            opt.step()
            opt.zero_grad()
    return [p.detach().clone() for p in net.parameters()]


def _worker(rank, world, port, avg, steps, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import parallel
    assert parallel.init_distributed("gloo")
    assert parallel.world_size() == world and parallel.rank() == rank
    net = _model()
    opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9)
    flat = parallel.FlatGrads(net.parameters())
    local = parallel.split_accumulation(avg, world)
    data = _batches(avg * steps)
    counter = 0
    for step in range(steps):
        mine = data[step * avg:(step + 1) * avg][rank::world]  # this rank's share of the step's micro-batches
        assert len(mine) == local
        for x, y in mine:
            (_loss(net(x), y) / avg).backward()   # accumulates into the flat buffer views
            counter += 1
            if counter % local == 0:
                flat.all_reduce()
                opt.step()
                flat.zero()
    lo, hi = flat.flat.data_ptr(), flat.flat.data_ptr() + flat.flat.numel() * 4
    for p in net.parameters():  # gradients really live in the flat buffer
        assert lo <= p.grad.data_ptr() < hi
    torch.save([p.detach().clone() for p in net.parameters()], os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_update_equals_single_process(tmp_path):
    import subprocess
    avg, steps, world = 4, 3, 2
    ref = _reference(avg, steps)
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), str(world), str(port), str(avg),
                               str(steps), str(tmp_path)]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    results = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(world)]
    for r in range(world):
        for a, b in zip(results[r], ref):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), f"rank {r} diverged from the single-process update"
    for a, b in zip(results[0], results[1]):
        assert torch.equal(a, b)  # replicas stay bit-identical


if __name__ == "__main__":
    r, w, port, avg, steps = (int(v) for v in sys.argv[1:6])
    _worker(r, w, port, avg, steps, sys.argv[6])
