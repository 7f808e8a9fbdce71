"""Multi-GPU support for the fine-tune loop: one process per GPU, torch.distributed over RCCL
("nccl" backend on ROCm) on xGMI, gloo on CPU for tests.

The reference has no collective at all: its only multi-device mechanism is launching
independent processes that shard the sequence list (src/train_online.py:184-186).  Two modes here:

* replicas: each rank fine-tunes its own sequences, no data-path collective (``shard_sequences``).
* data parallel inside one sequence (``FlatGrads.all_reduce``): the ``avg_grad_every_n`` accumulation
  micro-batches are spread over the ranks and the gradients are SUM-all-reduced once per optimizer
  step; with ``world * local_accum == avg_grad_every_n`` this reproduces the single-process update
  (src/train_online.py:92-101) up to fp32 summation order.

Gradients live in ONE flat fp32 buffer (each ``p.grad`` is a view), so zeroing is one memset and a collective
covers a contiguous slice.  The all-reduce is BUCKETED in the order the backward pass finishes gradients (stage 5
first): ``FlatGrads.all_reduce_begin`` issues one asynchronous all-reduce per bucket, each waiting only for ITS
gradients (the native backward publishes an event per bucket, ``fosvos_vgg_grad_bucket_wait``), so the large
stage-5 / stage-4 transfers run under the rest of the backward pass; ``all_reduce_finish`` joins them in front of
the optimizer step.  Frozen tensors (the bilinear deconvs, lr 0, no gradient is ever produced) are not sent.

Offline training over a split batch also needs the class counts of the WHOLE batch (the reference balances classes
over the batch tensor, src/layers/osvos_layers.py:28-39): ``batch_label_counts``.
"""
from __future__ import annotations

import os
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


SINGLE_RANK_ENV = "FOSVOS_DP_SINGLE_RANK"  # test hook, see collectives_on()


def init_distributed(backend: Optional[str] = None) -> bool:
    """Initialise the default process group from the torchrun environment; False when single-process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and os.environ.get(SINGLE_RANK_ENV) != "1":
        return False
    if dist.is_initialized():
        return True
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group(backend=backend)
    return True


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def collectives_on() -> bool:
    """Do the loops run their gradient collectives?  More than one rank - or, as a test hook (FOSVOS_DP_SINGLE_RANK=1 with an
    initialised process group), a single rank: the all-reduces then run over one rank and change no value, but every
    stream wait, event and asynchronous work handle of the real backend is exercised.  RCCL needs one device per rank, so
    this is how the RCCL choreography of the data-parallel step is run on a one-GPU box (tests/test_gpu_parallel.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get(SINGLE_RANK_ENV) == "1"


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_sequences(sequences: Sequence[str], group: Optional[int], group_size: Optional[int]) -> List[str]:
    """The reference's -sg/-sgs sharding: sequence i goes to group i % group_size
    (src/train_online.py:178-186)."""
    if group is None:
        return list(sequences)
    return [s for i, s in enumerate(sequences) if i % group_size == group]


def split_accumulation(avg_grad_every_n: int, world: int) -> int:
    """Micro-batches each rank runs per optimizer step so that world * local == avg_grad_every_n."""
    if avg_grad_every_n % world != 0:
        raise ValueError(f"avg_grad_every_n={avg_grad_every_n} must be a multiple of the world size {world} "
                         f"for the data-parallel update to equal the single-process one")
    return avg_grad_every_n // world


# Gradient buckets in the order the backward pass completes them, by parameter-name prefix (state_dict names of
# OSVOS_VGG).  Bucket index = the `bucket` argument of fosvos_vgg_grad_bucket_wait.
VGG_BUCKETS = (("stages.4.",), ("stages.3.",), ("stages.2.",), ("stages.0.", "stages.1."),
               ("side_prep.", "score_dsn.", "fuse."))
VGG_EARLY_BUCKETS = 3  # buckets 0-2 are published while the backward pass is still running (97 % of the bytes)


def batch_label_counts(gts: torch.Tensor) -> Optional[torch.Tensor]:
    """float64 [2] {positives, pixels} of the data-parallel batch `gts` is this rank's shard of (SUM over ranks);
    None in a single process (the loss then counts its own tensor, as the reference does)."""
    if world_size() == 1:
        return None
    counts = torch.stack([(gts >= 0.5).sum().to(torch.float64),
                          torch.tensor(float(gts.numel()), dtype=torch.float64, device=gts.device)])
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


class FlatGrads:
    """All trainable gradients as views of one flat fp32 buffer."""

    def __init__(self, params: Iterable[torch.nn.Parameter], names: Optional[Sequence[str]] = None,
                 buckets: Sequence[Sequence[str]] = VGG_BUCKETS, frozen: Sequence[str] = ("upscale.", "upscale_.")):
        params = list(params)
        names = list(names) if names is not None else None
        if names is not None and len(names) != len(params):
            raise ValueError("FlatGrads: names and params differ in length")
        # re-use the buffer a previous loop already attached to these parameters
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGrads: no trainable parameters")
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        # 16-byte aligned slices keep the vector paths of the SGD kernel
        offs, cur = [], 0
        for p in self.params:
            offs.append(cur)
            cur += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(cur, dtype=torch.float32, device=dev)
        self.numel = total
        self._offsets = list(offs)
        for p, o in zip(self.params, offs):
            p.grad = self.flat[o:o + p.numel()].view_as(p)
        self._views = [p.grad for p in self.params]  # (identity of these objects = "nobody re-pointed a .grad since")
        self.cache: dict = {}  # small per-loop constants of the training loops (device scalars, pinned staging), keyed by them
        # bucket b = one contiguous slice [lo, hi) of the flat buffer; without names: one bucket, everything.
        # Trainable tensors outside every bucket form trailing buckets of their own, except those under `frozen`:
        # the engine never produces a gradient for them (the reference keeps them at lr 0), their slice stays zero.
        self.slices: List[Tuple[int, int]] = [(0, cur)]
        self.bucket_params: List[List[torch.nn.Parameter]] = [list(self.params)]  # the parameters inside each slice
        # slice b holds bucket `bucket_ids[b]` of `buckets` (the index fosvos_vgg_grad_bucket_wait takes): a bucket without
        # trainable parameters leaves no slice, so slice and bucket numbers differ as soon as a stage is frozen; slices of
        # tensors outside every bucket (and the single slice of the unnamed form) wait for the pass's LAST bucket
        self.bucket_ids: List[int] = [len(buckets) - 1]
        if names is not None:
            trainable = [n for n, p in zip(names, params) if p.requires_grad]
            ends = offs[1:] + [cur]
            self.slices = []
            self.bucket_params = []
            self.bucket_ids = []
            covered = set()
            for bucket_id, prefixes in enumerate(buckets):
                idx = [i for i, n in enumerate(trainable) if n.startswith(tuple(prefixes))]
                if not idx:
                    continue
                if idx != list(range(idx[0], idx[-1] + 1)):
                    raise ValueError(f"FlatGrads: bucket {prefixes} is not contiguous in parameter order")
                self.slices.append((offs[idx[0]], ends[idx[-1]]))
                self.bucket_params.append([self.params[i] for i in idx])
                self.bucket_ids.append(bucket_id)
                covered.update(idx)
            run: List[int] = []
            for i, n in enumerate(trainable + [None]):
                if n is not None and i not in covered and not n.startswith(tuple(frozen)):
                    run.append(i)
                elif run:
                    self.slices.append((offs[run[0]], ends[run[-1]]))
                    self.bucket_params.append([self.params[i] for i in run])
                    self.bucket_ids.append(len(buckets) - 1)
                    run = []
        self._works: list = []

    @classmethod
    def attach(cls, net, params: Iterable[torch.nn.Parameter], names: Optional[Sequence[str]] = None) -> "FlatGrads":
        """The flat buffer of `net`: the one a previous loop attached, zeroed, while its parameters still point into it -
        otherwise a new one.  (A new buffer per call of a training loop costs a 60 MB device allocation every second call:
        the parameters keep the previous buffer alive until their `.grad` is re-pointed.)"""
        params = list(params)
        old = getattr(net, "_fosvos_flat_grads", None)
        if old is not None and old.still_attached([p for p in params if p.requires_grad]):
            old.zero()
            return old
        new = cls(params, names=names)
        try:
            net._fosvos_flat_grads = new
        except Exception:  # a module that refuses attributes: just do without the cache
            pass
        return new

    def still_attached(self, params: Sequence[torch.nn.Parameter]) -> bool:
        if len(params) != len(self.params) or any(a is not b for a, b in zip(params, self.params)):
            return False
        if all(p.grad is v for p, v in zip(self.params, self._views)):  # the very view objects this buffer handed out
            return True
        base, esz = self.flat.data_ptr(), self.flat.element_size()
        return all(p.grad is not None and p.grad.data_ptr() == base + o * esz and p.grad.numel() == p.numel()
                   for p, o in zip(self.params, self._offsets))

    @classmethod
    def attach_module(cls, net) -> "FlatGrads":
        """`attach` for all parameters of `net`, named: the loops' entry.  With a buffer already attached the names are not
        even listed (a training call of a few steps pays for every microsecond in front of its first kernel)."""
        old = getattr(net, "_fosvos_flat_grads", None)
        if old is not None and old.still_attached([p for p in net.parameters() if p.requires_grad]):
            old.zero()
            return old
        named = list(net.named_parameters())
        return cls.attach(net, [p for _, p in named], names=[n for n, _ in named])

    def zero(self, buckets: Optional[Sequence[int]] = None) -> None:
        """Zero the whole buffer, or only the slices of the given buckets."""
        if buckets is None:
            self.flat.zero_()
            return
        spans = sorted(self.slices[b] for b in buckets)
        merged: List[List[int]] = []
        for lo, hi in spans:  # adjacent slices share one memset (stages 5-3 are one run of the buffer)
            if merged and merged[-1][1] == lo:
                merged[-1][1] = hi
            else:
                merged.append([lo, hi])
        for lo, hi in merged:
            self.flat[lo:hi].zero_()

    def all_reduce(self, async_op: bool = False):
        """SUM of the whole buffer over ranks in one collective (no-op in a single process)."""
        if not collectives_on():
            return None
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)

    def all_reduce_begin(self, wait_bucket: Optional[Callable[[int], None]] = None,
                         on_bucket: Optional[Callable[[int, object], None]] = None) -> None:
        """One asynchronous SUM all-reduce per slice, in completion order.  `wait_bucket(id)` is called right before a
        slice's collective is enqueued with the slice's NATIVE bucket id (`bucket_ids`) and must make the CURRENT stream
        wait for that bucket's gradients (on the HIP path: fosvos_vgg_grad_bucket_wait on the communication stream);
        None = the current stream already follows the whole backward pass."""
        if not collectives_on():
            return
        for b, (lo, hi) in enumerate(self.slices):
            if wait_bucket is not None:
                wait_bucket(self.bucket_ids[b])
            if on_bucket is not None:
                on_bucket(b, None)  # (timing: an event in front of the collective)
            self._works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
            if on_bucket is not None:
                on_bucket(b, self._works[-1])

    def all_reduce_wait(self, bucket: int) -> None:
        """Make the current stream (the host, for CPU tensors) wait for bucket `bucket`'s collective of all_reduce_begin
        only - what lets a loop step the parameters of an early bucket while the later ones are still on the wire."""
        if bucket < len(self._works) and self._works[bucket] is not None:
            self._works[bucket].wait()
            self._works[bucket] = None

    def all_reduce_finish(self) -> None:
        """Make the current stream (the host, for CPU tensors) wait for every collective of all_reduce_begin."""
        for w in self._works:
            if w is not None:
                w.wait()
        self._works = []


COMM_TIMING = False  # bench.py sets it for its data-parallel runs


class GradSync:
    """The gradient exchange of one optimizer step in the data-parallel loops (train_online / train_offline):

        sync.arm()                  before the LAST backward of the accumulation cycle
        loss.backward(...)
        sync.begin()                right after it: bucketed asynchronous all-reduce on a communication stream
        ...
        sync.finish()               in front of optimizer.step(): the current stream waits for all of it

    On the GPU each bucket's collective waits only for that bucket's gradients (``net.wait_grad_bucket``), so the
    stage-5, -4 and -3 transfers (97 % of the bytes) run under the rest of the backward pass.  With CPU tensors (gloo
    tests) the same calls run the same bucketed arithmetic without streams."""

    def __init__(self, net, flat: "FlatGrads"):
        self.net, self.flat = net, flat
        self.active = collectives_on()
        self._comm = None
        # Communication timing (COMM_TIMING / FOSVOS_COMM_TIMING=1; GPU only): per optimizer step, events on the
        # communication stream in front of and behind every bucket's all-reduce, an event on the main stream where the data-
        # gradient chain of the cycle's last backward pass ended, and one on the auxiliary stream where its weight-gradient
        # kernels ended (the later of the two = when the optimizer step could start without communication): what
        # timing_summary() turns into per-bucket offsets and the exposed communication time per step.
        self.timing = COMM_TIMING or os.environ.get("FOSVOS_COMM_TIMING", "0") == "1"
        self._steps: list = []   # per step: {"dgrad_end": ev, "aux_end": ev, "ready": ev, "buckets": [(start ev, end ev, bytes)]}

    def arm(self) -> None:
        if self.active and hasattr(self.net, "publish_grad_buckets"):
            self.net.publish_grad_buckets = True

    def begin(self) -> None:
        if not self.active:
            return
        net = self.net
        if hasattr(net, "publish_grad_buckets"):
            net.publish_grad_buckets = False
        # the bucket events exist only where the native backward pass recorded them (not under FOSVOS_PY_ENGINE=1)
        if self.flat.flat.is_cuda and hasattr(net, "wait_grad_bucket") and getattr(net, "publishes_grad_buckets", True):
            from fosvos_hip import engine
            d = self.flat.flat.device
            if self._comm is None:  # one communication stream per device and process (engine.shared_stream)
                self._comm = engine.shared_stream(d.index if d.index is not None else torch.cuda.current_device(), "comm")
            main = torch.cuda.current_stream(self.flat.flat.device)
            comm = self._comm

            def wait(b: int) -> None:
                net.wait_grad_bucket(b, comm)
                if b >= VGG_EARLY_BUCKETS:  # the tail buckets also hold gradients the host side adds on the main stream (score_dsn)
                    comm.wait_stream(main)

            on_bucket = None
            if self.timing:
                rec = {"dgrad_end": torch.cuda.Event(enable_timing=True), "aux_end": torch.cuda.Event(enable_timing=True),
                       "ready": None, "buckets": []}
                rec["dgrad_end"].record(main)  # (the backward call has returned: its data-gradient chain ends here on `main`)
                # ... and its weight-gradient kernels end here on the auxiliary stream: without communication the optimizer
                # step could start at the later of the two
                rec["aux_end"].record(engine.shared_stream(d.index if d.index is not None else torch.cuda.current_device(), "aux"))
                self._steps.append(rec)
                slices = self.flat.slices

                def on_bucket(b: int, work) -> None:
                    ev = torch.cuda.Event(enable_timing=True)
                    if work is None:
                        ev.record(comm)
                        rec["buckets"].append([ev, None, (slices[b][1] - slices[b][0]) * 4])
                    else:
                        work.wait()  # the communication stream itself waits for the collective, so that the event is behind it
                        ev.record(comm)
                        rec["buckets"][-1][1] = ev

            with torch.cuda.stream(comm):
                self.flat.all_reduce_begin(wait, on_bucket)
        elif self.flat.flat.is_cuda:  # no per-bucket events: the communication stream follows the whole backward pass
            if self._comm is None:
                self._comm = torch.cuda.Stream(device=self.flat.flat.device)
            if hasattr(net, "join_gradients"):
                net.join_gradients()
            self._comm.wait_stream(torch.cuda.current_stream(self.flat.flat.device))
            with torch.cuda.stream(self._comm):
                self.flat.all_reduce_begin(None)
        else:
            self.flat.all_reduce_begin(None)

    def wait_bucket(self, bucket: int) -> None:
        """The current stream waits for bucket `bucket`'s all-reduce (begun by begin()) only."""
        if self.active:
            self.flat.all_reduce_wait(bucket)

    def finish(self) -> None:
        if self.active:
            if self.timing and self._steps and self._steps[-1]["ready"] is None and self.flat.flat.is_cuda:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream(self.flat.flat.device))  # here the optimizer step could start
                self._steps[-1]["ready"] = ev
            self.flat.all_reduce_finish()

    def timing_summary(self) -> Optional[dict]:
        """After a device sync: per bucket the mean start / end of its all-reduce in ms after the end of the data-gradient
        chain, its bytes, `compute_done_ms_after_dgrad_end` (when the weight-gradient stream ended: the moment the optimizer
        step could have started without communication) and `comm_exposed_ms_per_step` = how long after THAT the last
        bucket's all-reduce ended (0 when the communication hid behind the rest of the backward pass)."""
        steps = [r for r in self._steps if r["ready"] is not None and r["buckets"] and all(b[1] is not None for b in r["buckets"])]
        if not steps:
            return None
        n_b = len(steps[0]["buckets"])
        start = [0.0] * n_b
        end = [0.0] * n_b
        exposed, done = 0.0, 0.0
        for r in steps:
            for i, (e0, e1, _) in enumerate(r["buckets"]):
                start[i] += r["dgrad_end"].elapsed_time(e0)
                end[i] += r["dgrad_end"].elapsed_time(e1)
            compute_done = max(0.0, r["dgrad_end"].elapsed_time(r["aux_end"]))
            done += compute_done
            exposed += max(0.0, r["dgrad_end"].elapsed_time(r["buckets"][-1][1]) - compute_done)
        n = len(steps)
        return {"optimizer_steps": n, "comm_exposed_ms_per_step": exposed / n, "compute_done_ms_after_dgrad_end": done / n,
                "buckets": [{"bytes": steps[0]["buckets"][i][2], "start_ms_after_dgrad_end": start[i] / n,
                             "end_ms_after_dgrad_end": end[i] / n} for i in range(n_b)],
                "note": "events on the communication stream around every bucket's all-reduce, relative to the end of the "
                        "data-gradient chain of the cycle's last backward pass (negative = under the backward pass)"}
