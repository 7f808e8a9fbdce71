"""One-shot online fine-tuning (reference: src/train_online.py).

Same entry points - ``train_and_test(net_provider, seq_name, settings)`` and ``_train(...)`` with the
reference's positional arguments - and the same loop semantics (src/train_online.py:70-107): forward,
class-balanced BCE on ``outputs[-1]`` with ``size_average=False``, ``loss /= avg_grad_every_n``,
backward, ``optimizer.step(); optimizer.zero_grad()`` every ``avg_grad_every_n``-th iteration.  The
arithmetic runs on the HIP kernels; the loss is accumulated on the device and only read back at the
reference's logging points, so the loop does not synchronise every iteration.

Run:  python train_online.py --synthetic --n-epochs 100        (one GPU)
      torchrun --nproc-per-node 8 train_online.py --synthetic --data-parallel --avg-grad-every-n 8
"""
import sys
import os
import timeit
from pathlib import Path
from typing import Optional

import torch
from torch import optim

from config.mypath import Path as P
from layers.osvos_layers import class_balanced_cross_entropy_loss, class_balanced_cross_entropy_loss_frames, stage_frames_loss
from util import gpu_handler, io_helper, experiment_helper, args_helper
from util.logger import get_logger
from util.network_provider import NetworkProvider, provider_mapping
from util.settings import OnlineSettings
import parallel

log = get_logger(__file__)
_hip_cbce = class_balanced_cross_entropy_loss  # tests may rebind the module-level name to a CPU stand-in

# run-wide locations; the reference keeps these as module globals set in __main__
save_dir_models = Path('models')
save_dir_results = Path('results')
path_stem = 'vgg16/online'
db_root_dir = None
synthetic_size = None       # (H, W) when --synthetic
data_parallel = False

sequences_val = ['blackswan', 'bmx-trees', 'breakdance', 'camel', 'car-roundabout', 'car-shadow', 'cows',
                 'dance-twirl', 'dog', 'drift-chicane', 'drift-straight', 'goat', 'horsejump-high', 'kite-surf',
                 'libby', 'motocross-jump', 'paragliding-launch', 'parkour', 'scooter-black', 'soapbox']


def train_and_test(net_provider: NetworkProvider, seq_name: str, settings: OnlineSettings) -> None:
    io_helper.write_settings(save_dir_models, net_provider.name, settings, variant_offline=settings.variant_offline,
                             variant_online=settings.variant_online)
    summary_writer = _get_summary_writer(path_stem)

    if settings.is_training:
        net_provider.load_network_train()
        data_loader = io_helper.get_data_loader_train(db_root_dir, settings.batch_size_train, seq_name,
                                                      synthetic=synthetic_size)
        optimizer = net_provider.get_optimizer()
        _train(net_provider, data_loader, optimizer, summary_writer, seq_name, settings.start_epoch, settings.n_epochs,
               settings.avg_grad_every_n, settings.snapshot_every_n)

    if settings.is_testing:
        if not settings.is_training:
            net_provider.load_network_test(sequence=seq_name)
        data_loader = io_helper.get_data_loader_test(db_root_dir, settings.batch_size_test, seq_name,
                                                     synthetic=synthetic_size)
        if settings.variant_offline is None:
            save_dir = save_dir_results / net_provider.name / 'online'
        else:
            save_dir = (save_dir_results / net_provider.name / str(settings.variant_offline) /
                        str(settings.variant_online))
        experiment_helper.test(net_provider, data_loader, save_dir, settings.is_visualizing_results,
                               settings.eval_speeds, seq_name=seq_name)


def _max_group() -> int:
    """Micro-batches of one accumulation cycle that may run as one batched pass (FOSVOS_MICROBATCH_GROUP, default 5 = the
    reference's whole cycle, avg_grad_every_n; 1 = the reference's one-by-one order).  A group never crosses an optimizer
    step, so the default runs a cycle of up to five same-size frames as ONE forward / backward pass: the fewest launches
    and the fullest kernels (measured on the 480x854 step: 1086 frames/s against 1010 with 3 + 2 and 932 with 2 + 2 + 1).
    Longer cycles are cut into groups of at most this many frames (activation memory grows with the group)."""
    try:
        return max(1, int(os.environ.get('FOSVOS_MICROBATCH_GROUP', '5')))
    except ValueError:
        return 5


def _group_window() -> int:
    """How many micro-batches of an accumulation cycle the loop looks at together before it forms its batched passes
    (FOSVOS_GROUP_WINDOW, default 16; never more than the cycle itself).  The reference's augmentation draws a random scale
    per iteration (src/dataloaders/custom_transforms.py:63-76), so consecutive frames rarely share a size; gradients inside a
    cycle are a sum, so the micro-batches of the window are bucketed BY SHAPE and every bucket runs as one batched pass."""
    try:
        return max(1, int(os.environ.get('FOSVOS_GROUP_WINDOW', '16')))
    except ValueError:
        return 16


def _losses_per_frame(fused, gts, backward_seed=None, staged=None):
    """[k] per-frame losses of a batched pass; one fused op on the GPU, the reference's function per slice elsewhere.
    staged: the loss was started beside the forward pass (osvos_layers.stage_frames_loss): its values are written by
    staged.finish(), which the caller queues behind the backward pass."""
    if fused.is_cuda and class_balanced_cross_entropy_loss is _hip_cbce:
        return class_balanced_cross_entropy_loss_frames(fused, gts, size_average=False, backward_seed=backward_seed,
                                                        staged=staged)
    return torch.stack([class_balanced_cross_entropy_loss(fused[i:i + 1], gts[i:i + 1], size_average=False)
                        for i in range(fused.shape[0])])


class _Landed:
    """Stand-in for a recorded CUDA event when the loop runs on CPU tensors."""

    def query(self):
        return True

    def synchronize(self):
        pass


def _get_summary_writer(seq_name: str):
    return io_helper.get_summary_writer(Path('tensorboard') / path_stem)


def _train(net_provider: NetworkProvider, dataloader, optimizer: optim.SGD, summary_writer, seq_name: str,
           start_epoch: int, n_epochs: int, avg_grad_every_n: int, snapshot_every_n: int) -> dict:
    log.info('Start of Online Training, sequence: ' + seq_name)
    net = net_provider.network
    net.accumulate_grads_in_place = True  # this loop only ever calls loss.backward()
    net.compute_side_outputs = False      # ... on outputs[-1] only (src/train_online.py:80): skip the 4 side logit maps
    # weights are constant inside an accumulation cycle: let the next forward overlap the wgrad tail of this backward
    net.defer_wgrad_join = os.environ.get('FOSVOS_DEFER_JOIN', '1') != '0'
    world = parallel.world_size() if data_parallel else 1
    dp_on = data_parallel and parallel.collectives_on()  # (world > 1, or the single-rank test hook)
    local_accum = parallel.split_accumulation(avg_grad_every_n, world)
    # gradients live in one flat fp32 buffer: the wgrad kernels accumulate straight into it, zeroing is one memset,
    # and under data parallelism it is the single all-reduce payload
    flat = parallel.FlatGrads.attach_module(net)
    sync = parallel.GradSync(net, flat)

    # on the GPU the optimizer step is split by gradient bucket (see run_group); FOSVOS_SPLIT_STEP=0 = one step
    # (slice indices of the flat buffer; flat.bucket_ids maps a slice to the native bucket its gradients are published as)
    early_buckets = [b for b, bid in enumerate(flat.bucket_ids) if bid < parallel.VGG_EARLY_BUCKETS]
    late_buckets = [b for b in range(len(flat.slices)) if b not in early_buckets]
    split_step = (flat.flat.is_cuda and hasattr(net, 'wait_grad_bucket') and bool(early_buckets) and bool(late_buckets)
                  and getattr(net, 'publishes_grad_buckets', True)  # the native backward pass records the bucket events
                  and hasattr(optimizer, '_tables') and os.environ.get('FOSVOS_SPLIT_STEP', '1') != '0')
    early_params = [p for b in early_buckets for p in flat.bucket_params[b]]
    early_ids = {id(p) for p in early_params}
    late_params = [p for group in optimizer.param_groups for p in group['params'] if id(p) not in early_ids]
    early_prefixes = tuple(pre for b in early_buckets for pre in parallel.VGG_BUCKETS[flat.bucket_ids[b]])

    # lab switch (A/B only): 0 = separate gradient memsets behind the optimizer step and an unannounced backward seed
    fuse_small = os.environ.get('FOSVOS_LOOP_FUSE', '1') != '0'
    # The loss of a batched pass in three stages (class counts in front of the forward pass, values and host copy behind the
    # backward pass; fosvos_cbce_loss_frames_parts): two launches and the copy leave the chain of small kernels between the
    # passes.  +0.5 % when that chain was 150 us long, neutral after the head kernels got shorter, +0.3 % on the final build
    # (5 of 5 interleaved rounds, profiles/r04_lab_step_ab_tunables.txt).  The same arithmetic either way (tested bit for
    # bit); FOSVOS_STAGE_LOSS=0 = the loss as one call between the passes.
    stage_losses = os.environ.get('FOSVOS_STAGE_LOSS', '1') == '1'
    # Gradient buffers without zeroing: a cycle that is ONE batched pass (the usual case: nAveGrad frames of one shape) WRITES
    # its gradients (net.overwrite_grads) instead of adding them to buffers the previous optimizer step had to zero - one
    # write and one read of every gradient less per cycle, in the HBM-bound tail of the cycle (+0.6 %,
    # profiles/r04_lab_step_ab_tunables.txt).  The optimizer step then leaves the gradients in place ("stale"), and a cycle
    # of several passes - which do add - zeroes the buffer first.  The same values either way (a sum that starts from zero):
    # tested bit for bit.  FOSVOS_GRAD_OVERWRITE=0: zero in the optimizer step, always add.
    lazy_zero = (fuse_small and flat.flat.is_cuda and hasattr(net, 'overwrite_grads')
                 and os.environ.get('FOSVOS_GRAD_OVERWRITE', '1') != '0')
    grads_stale = [False]
    # A cycle whose micro-batches cannot run as ONE batched pass (frames of different sizes - the reference's augmentation
    # draws a new scale per iteration - or FOSVOS_MICROBATCH_GROUP < nAveGrad) runs its passes on two alternating streams:
    # the weights do not change inside a cycle and every pass has its own arena, so the forward pass of one micro-batch may
    # run beside the backward pass of the previous one (their weight-gradient kernels share one stream and stay in order, so
    # the accumulation into the gradients does too).  FOSVOS_PASS_STREAMS=0: one stream.
    pass_streams = None
    if flat.flat.is_cuda:
        from fosvos_hip import engine as _engine
        dev_index = flat.flat.device.index if flat.flat.device.index is not None else torch.cuda.current_device()
        # the library's auxiliary streams exist from here on, in their fixed creation order (engine.shared_stream)
        _engine.shared_stream(dev_index, "comm")
        if (hasattr(net, 'join_gradients') and getattr(net, 'defer_wgrad_join', False)
                and os.environ.get('FOSVOS_PASS_STREAMS', '1') != '0'):
            pass_streams = [None, _engine.shared_stream(dev_index, "pass")]  # None = the caller's stream
    n_samples = len(dataloader)
    loss_tr = []
    counter_gradient = 0
    log_every = max(n_epochs // 20, 1)  # the reference divides by n_epochs // 20, which is 0 below 20 epochs
    device = next(net.parameters()).device
    max_group = _max_group()
    consts = flat.cache.get(('online', avg_grad_every_n, max_group, str(device)))
    if consts is None:  # (constants of the loop, kept with the gradient buffer: a short call should not re-create them)
        consts = flat.cache[('online', avg_grad_every_n, max_group, str(device))] = {
            'inv_avg': torch.ones((), device=device) / avg_grad_every_n,
            'inv_avg_k': torch.full((max_group,), 1.0 / avg_grad_every_n, device=device),  # backward seed of a batched pass
            'ring': torch.empty((64, max_group), dtype=torch.float32).pin_memory() if device.type == 'cuda' else None}
    inv_avg, inv_avg_k = consts['inv_avg'], consts['inv_avg_k']

    # The reference reads loss.item() every iteration and the running loss at 20 logging points per run (device->host
    # syncs that drain the launch queue).  Here every pass sends its per-frame losses to pinned memory with ONE asynchronous
    # copy and the host does the bookkeeping (running sum, logging points) once the copy has landed: no device-side
    # accumulator kernels, no sync.  `loss_tr` fills in iteration order, a few passes behind the device.
    pending_logs = []   # per window: ([(epoch, minibatch index, host tensor, position)] in iteration order, events, ring slots)
    ring, ring_free = None, []
    if device.type == 'cuda':
        ring = consts['ring']  # (every slot is free again: the previous call ended with a device sync)
        ring_free = list(range(ring.shape[0]))
    running_host = [0.0]

    def flush_logs(block: bool) -> None:
        while pending_logs:
            items, landed, slots = pending_logs[0]
            # every pass's copy has its own event: the passes of a window alternate between two streams, so the last
            # event alone says nothing about the copies queued on the other stream
            if block:
                for ev in landed:
                    ev.synchronize()
            elif not all(ev.query() for ev in landed):
                break
            pending_logs.pop(0)
            for ep, mb, host_vals, i in items:  # iteration order of the reference's loop, whatever order the passes ran in
                running_host[0] += float(host_vals[i])
                if is_log_epoch(ep):
                    value = running_host[0] / n_samples
                    running_host[0] = 0.0
                    loss_tr.append(value)
                    log.info('[Epoch {0}: {1}, numImages: {2}]'.format(seq_name, ep + 1, mb + 1))
                    log.info('Loss {0}: {1}'.format(seq_name, value))
                    summary_writer.add_scalar('data/total_loss_epoch', value, ep)
            ring_free.extend(slots)

    window_logs = []  # (frames, host tensor, event, ring slot) of the passes of the window being run

    def record_losses(group, losses) -> None:
        """losses: [k] detached device tensor, frame by frame in the order of `group`."""
        frames = [(g[0], g[1]) for g in group]
        if ring is not None:
            if not ring_free:
                flush_logs(True)
            if not ring_free:  # every slot is held by the window in flight: a plain (pageable, synchronous) copy instead
                window_logs.append((frames, losses.to('cpu'), _Landed(), None))
                return
            slot = ring_free.pop()
            host_vals = ring[slot, :len(frames)]
            host_vals.copy_(losses, non_blocking=True)
            landed = torch.cuda.Event()
            landed.record()
        else:  # CPU tensors (the gloo tests of the data-parallel wiring): nothing to wait for
            slot, host_vals, landed = None, losses.clone(), _Landed()
        window_logs.append((frames, host_vals, landed, slot))

    def close_window_logs() -> None:
        """The passes of a window ran bucket by bucket; the bookkeeping (running loss, logging points) follows the
        reference's iteration order: one pending entry per window, sorted, waiting on the copies of ALL its passes."""
        if not window_logs:
            return
        items = sorted(((ep, mb, host_vals, i) for frames, host_vals, _, _ in window_logs
                        for i, (ep, mb) in enumerate(frames)), key=lambda t: (t[0], t[1]))
        slots = [slot for _, _, _, slot in window_logs if slot is not None]
        pending_logs.append((items, [landed for _, _, landed, _ in window_logs], slots))
        window_logs.clear()
        flush_logs(False)

    def is_log_epoch(epoch: int) -> bool:
        return epoch % log_every == log_every - 1

    def is_snapshot_epoch(epoch: int) -> bool:
        return (epoch % snapshot_every_n) == snapshot_every_n - 1

    side_stream_used = [False]  # a pass of the current cycle ran on pass_streams[1]: the optimizer step waits for it
    last_pass_on_side = [False]  # ... and it was the most recent pass

    def run_group(group, stream=None) -> None:
        """One forward / loss / backward pass over the micro-batches of `group` (iterations of the reference's loop with
        the same frame size, inside one accumulation cycle; consecutive or not - see run_window).  The weights do not change inside a cycle, so running k
        iterations as one batch of k frames leaves every frame's logits, loss (the class weights are still counted per
        frame) and gradient contribution what the one-by-one loop computes; only the order of the fp32 sums over frames
        differs.  What it buys on the GPU: k frames per kernel launch (the small stage-5 layers fill the chip without
        split-K, fewer launch gaps) and one set of weight-gradient partial slabs per group instead of per frame."""
        nonlocal counter_gradient, n_iters
        k = len(group)
        last_pass_on_side[0] = stream is not None
        if stream is not None:
            side_stream_used[0] = True
            with torch.cuda.stream(stream):
                run_pass(group, k)
        else:
            run_pass(group, k)
        counter_gradient += k
        n_iters += k
        close_cycle_if_due()

    def run_pass(group, k) -> None:
        """Forward, loss and backward pass of one group, on the current stream."""
        if k == 1:
            inputs, gts = group[0][2]['image'], group[0][2]['gt']
        else:
            inputs = torch.cat([g[2]['image'] for g in group])
            gts = torch.cat([g[2]['gt'] for g in group])
        inputs, gts = gpu_handler.cast_cuda_if_possible([inputs, gts])

        # A batched pass spreads its loss over three places: the class counts of the labels are taken here, in front of the
        # forward pass (they need no logits), the loss kernel proper sits between the passes, and the loss VALUES - which
        # only the log reads - are finished and copied to the host behind the backward pass.  Between the passes the
        # whole chip waits on a chain of small kernels: this takes two of them and the copy out of that chain.
        staged = None
        if k > 1 and stage_losses and gts.is_cuda and class_balanced_cross_entropy_loss is _hip_cbce:
            staged = stage_frames_loss(gts)

        outputs = net.forward(inputs)

        # (the seed of the backward pass below is announced to the loss: its kernel writes the gradient times 1 / nAveGrad
        # and the loss's own backward has nothing left to multiply)
        seeded = fuse_small and outputs[-1].is_cuda and class_balanced_cross_entropy_loss is _hip_cbce
        if k == 1:
            loss = (class_balanced_cross_entropy_loss(outputs[-1], gts, size_average=False,
                                                      backward_seed=(inv_avg, 1.0 / avg_grad_every_n))
                    if seeded else class_balanced_cross_entropy_loss(outputs[-1], gts, size_average=False))
            losses = loss.detach().reshape(1)
        else:
            # [k]; the sum over frames is taken by the backward seed
            loss = _losses_per_frame(outputs[-1], gts, (inv_avg_k, 1.0 / avg_grad_every_n) if seeded else None, staged)
            losses = loss.detach()
        if staged is None:
            record_losses(group, losses)

        # reference: `loss /= nAveGrad; loss.backward()` (src/train_online.py:92-93).  Seeding the backward pass with
        # 1/nAveGrad is the same gradient (the division's own backward produces exactly this factor) without the
        # three tiny kernels of the division, the ones-fill and its backward on the critical path
        closes_cycle = (counter_gradient + k) % local_accum == 0
        last_of_cycle = dp_on and closes_cycle
        if hasattr(net, 'last_pass_of_cycle'):
            net.last_pass_of_cycle = closes_cycle
        if last_of_cycle:
            sync.arm()
        if split_step and closes_cycle:
            net.publish_grad_buckets = True
        loss.backward(inv_avg if k == 1 else inv_avg_k[:k])
        if last_of_cycle:
            sync.begin()
        if staged is not None:
            # the loss VALUES: one small kernel and the copy to the host, behind the backward pass.  (Queued on the
            # communication stream instead, behind an event, they cost 0.6 % - a third stream with work in flight - where on
            # this stream they cost nothing measurable: profiles/r04_lab_step_ab_head_loss.txt.)
            staged.finish()
            record_losses(group, losses)
        # (the reference also sums loss.item() into a per-epoch tensorboard scalar, src/train_online.py:94-104: one
        # device sync per frame; the per-pass asynchronous copy above carries the same information without it)

    def wait_side_stream() -> None:
        """The caller's stream waits for the passes of this cycle that ran on the second pass stream.  (The EARLY part of a
        split optimizer step does not need this for the EARLIER passes: the weight-gradient kernels of all passes share one
        stream, in order, and a pass's last weight-gradient kernels wait for the end of its data-gradient chain, so the
        bucket events of the cycle's last pass also say that every earlier pass has left the stages they cover.  The closing
        pass's OWN data-gradient chain is not covered by its bucket events - a stage's event is recorded behind the weight
        gradient of the stage's first conv, which is issued in front of that conv's data gradient - so the closing pass runs
        on the caller's stream, where the step queues behind it; see run_window and close_cycle_if_due.)"""
        if side_stream_used[0]:
            torch.cuda.current_stream(device).wait_stream(pass_streams[1])
            side_stream_used[0] = False

    def close_cycle_if_due() -> None:
        """The optimizer step behind the cycle's last pass, on the caller's stream."""
        nonlocal counter_gradient
        if counter_gradient % local_accum == 0:
            if split_step:
                # Stages 5-3 hold 97 % of the parameters and their gradients are final well before the backward pass ends.
                # Their share of the optimizer step, the zeroing of their gradients and the repacking of their weights are
                # queued on the main stream right here - behind the data-gradient chain, while the weight-gradient stream
                # is still working through stages 3-1 - and only the small rest waits for that stream.
                # Data parallel: "final" means all-reduced - the early buckets' collectives were begun while the backward pass
                # was still running, and each is waited for on its own.
                net.publish_grad_buckets = False
                if last_pass_on_side[0]:
                    # (run_window schedules a window's last pass on the caller's stream, so this is a guard: the early step
                    # rewrites and repacks weights the closing pass's data-gradient chain on the other stream still reads)
                    wait_side_stream()
                for b in early_buckets:
                    if dp_on:
                        sync.wait_bucket(b)
                    else:
                        net.wait_grad_bucket(flat.bucket_ids[b])
                optimizer.step(only=early_params, tag='early', zero_grad=fuse_small and not lazy_zero)  # (the step kernel zeroes what it read)
                if not fuse_small:
                    flat.zero(early_buckets)
                net.prepack_weights(early_prefixes)
                wait_side_stream()
                net.join_gradients()
                sync.finish()
                optimizer.step(only=late_params, tag='late', zero_grad=fuse_small and not lazy_zero)
                if not fuse_small:
                    flat.zero(late_buckets)
            else:
                wait_side_stream()
                net.join_gradients()
                sync.finish()  # the bucketed all-reduce begun right behind the cycle's last backward pass
                optimizer.step()
                if not lazy_zero:
                    flat.zero()
            grads_stale[0] = lazy_zero
            counter_gradient = 0

    def run_window(window) -> None:
        """The micro-batches the loop has collected (all inside one accumulation cycle, in iteration order) as batched
        passes: bucketed by frame shape in order of first appearance, each bucket cut into groups of at most `max_group`
        frames.  The cycle's gradient is a sum over its micro-batches, so the order of the passes changes nothing but the
        order of fp32 additions - and a loader that yields the same frames already sorted by shape runs the IDENTICAL
        passes (tested bit for bit).  Minibatches that are batches themselves (N > 1) run alone."""
        buckets = {}
        for item in window:
            shape = tuple(item[2]['image'].shape)
            buckets.setdefault(shape if shape[0] == 1 else ('batch', id(item)), []).append(item)
        groups = [items[i:i + max_group] for items in buckets.values() for i in range(0, len(items), max_group)]
        # (measured at 480x854, passes of 1 / 2 / 3+2 frames: +9.4 % / +2.0 % / -2.6 % - a pass of three or more frames does
        # better with its two forward chains side by side, which needs the auxiliary stream the previous pass's weight
        # gradients would still occupy)
        multi = pass_streams is not None and not dp_on and len(groups) > 1 and max(len(g) for g in groups) <= 2
        if lazy_zero:
            # a window that opens a cycle: one pass that is the whole cycle writes its gradients; anything else adds - to
            # zeros (queued here, on the caller's stream, in front of everything the window's passes do on either stream)
            whole_cycle = counter_gradient == 0 and len(groups) == 1 and len(groups[0]) == local_accum
            net.overwrite_grads = whole_cycle
            if counter_gradient == 0:
                if grads_stale[0] and not whole_cycle:
                    flat.zero()
                grads_stale[0] = False
        if multi:
            # both pass streams must see the last optimizer step and the repacked weight images: pack once here, on the
            # caller's stream, instead of inside the first forward pass (which the second stream would have to wait for)
            net.prepack_weights()
            pass_streams[1].wait_stream(torch.cuda.current_stream(device))
        if hasattr(net, 'forward_one_stream'):
            net.forward_one_stream = multi
        for i, group in enumerate(groups):
            # the window's LAST pass (the one that may close the cycle) always runs on the caller's stream: the split
            # optimizer step is queued there, behind that pass's data-gradient chain
            run_group(group, pass_streams[(len(groups) - 1 - i) % 2] if multi else None)
        if multi:
            # the window's minibatches were allocated on the caller's stream and read on the other one: before they are
            # released (window = [] below; a window can end without a cycle close) the caller's stream waits for it
            wait_side_stream()
        close_window_logs()
        epoch, _idx, _mb, end_of_epoch = window[-1]
        if end_of_epoch and is_snapshot_epoch(epoch) and parallel.rank() == 0:
            net_provider.save_model(epoch, sequence=seq_name)

    time_all_start = timeit.default_timer()
    n_iters = 0
    max_group = _max_group()
    max_window = max(max_group, _group_window())
    if hasattr(net, 'reserve_arena_frames'):
        net.reserve_arena_frames = min(max_group, avg_grad_every_n)  # (a pass never holds more frames than a cycle)
    window = []  # pending (epoch, minibatch index, minibatch, last of its epoch) tuples, all of one accumulation cycle
    for epoch in range(start_epoch, n_epochs):
        n_mb = len(dataloader)
        for minibatch_index, minibatch in enumerate(dataloader):
            end_of_epoch = minibatch_index == n_mb - 1
            window.append((epoch, minibatch_index, minibatch, end_of_epoch))
            # a window ends with its accumulation cycle, and before a snapshot is due (the snapshot must hold exactly the
            # updates up to its epoch)
            closes_cycle = (counter_gradient + len(window)) % local_accum == 0
            if closes_cycle or len(window) >= max_window or (end_of_epoch and is_snapshot_epoch(epoch)):
                run_window(window)
                window = []
    if window:
        run_window(window)

    net.defer_wgrad_join = False  # joins
    if lazy_zero:
        net.overwrite_grads = False
        if grads_stale[0]:  # leave the buffers as optimizer.zero_grad() would (src/train_online.py:100-104)
            flat.zero()
    if hasattr(net, 'forward_one_stream'):
        net.forward_one_stream = False
    net.compute_side_outputs = True
    time_enqueued = timeit.default_timer() - time_all_start  # how far ahead of the device the host loop ran
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    flush_logs(True)
    time_for_all = timeit.default_timer() - time_all_start
    n_images = len(dataloader)
    log.info('Train {0}: total time {1} sec'.format(seq_name, str(time_for_all)))
    log.info('Train {0}: {1} images'.format(seq_name, str(n_images)))
    log.info('Train {0}: time per sample {1} sec'.format(seq_name, str(time_for_all / max(n_images, 1))))
    return {'loss': loss_tr, 'seconds': time_for_all, 'iterations': n_iters, 'seconds_host_enqueue': time_enqueued,
            'comm_timing': sync.timing_summary() if sync.active and sync.timing else None}


def main(argv=None):
    global db_root_dir, synthetic_size, data_parallel, save_dir_models, save_dir_results
    args = args_helper.parse_args(is_online=True, argv=argv)
    if args.network != 'vgg16':
        raise SystemExit('only --network vgg16 is implemented on the HIP path (ResNet family: SURVEY.md §8 f4)')
    data_parallel = bool(args.data_parallel) and parallel.init_distributed()
    gpu_handler.select_gpu(args.gpu_id)

    db_root_dir = P.db_root_dir()
    synthetic_size = (args.height, args.width) if args.synthetic else None
    save_dir_models.mkdir(parents=True, exist_ok=True)
    save_dir_results.mkdir(parents=True, exist_ok=True)
    path_input_model = Path(args.parent_model) if args.parent_model else save_dir_models / 'vgg16_epoch-239.pth'
    path_output_model_base = save_dir_models / path_stem
    path_output_model_base.mkdir(parents=True, exist_ok=True)

    settings = OnlineSettings(is_training=args.is_training, is_testing=args.is_testing, start_epoch=0,
                              n_epochs=args.n_epochs or 10000, avg_grad_every_n=args.avg_grad_every_n or 5,
                              snapshot_every_n=args.n_epochs or 10000, is_testing_while_training=False, test_every_n=5,
                              batch_size_train=1, batch_size_test=1, is_visualizing_network=False,
                              is_visualizing_results=False, offline_epoch=240, variant_offline=args.variant_offline,
                              variant_online=args.variant_online, eval_speeds=args.eval_speeds)

    provider_class = provider_mapping[('online', args.network)]
    net_provider = provider_class(name=args.network, save_dir=(path_input_model, path_output_model_base),
                                  settings=settings, variant_offline=args.variant_offline,
                                  variant_online=args.variant_online)
    if args.synthetic and not path_input_model.exists():
        # no parent checkpoint offline: write a seeded random-init one so the load path is exercised
        from networks.osvos_vgg import OSVOS_VGG
        torch.manual_seed(0)
        if parallel.rank() == 0:
            path_input_model.parent.mkdir(parents=True, exist_ok=True)
            torch.save(OSVOS_VGG(pretrained=0).state_dict(), str(path_input_model))
        if data_parallel:
            torch.distributed.barrier()

    if args.sequence_name is None:
        # replicas: the reference's -sg/-sgs sharding; under torchrun without --data-parallel the ranks shard
        group, group_size = args.sequence_group, args.sequence_group_size
        if group is None and not data_parallel and parallel.init_distributed():
            group, group_size = parallel.rank(), parallel.world_size()
        sequences = parallel.shard_sequences(sequences_val, group, group_size)
        [train_and_test(net_provider, s, settings) for s in sequences]
    else:
        train_and_test(net_provider, args.sequence_name, settings)


if __name__ == '__main__':
    main()
