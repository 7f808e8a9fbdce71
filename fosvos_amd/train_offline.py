"""Parent-network (offline) training (reference: src/train_offline.py).

Same entry points and loop semantics (src/train_offline.py:77-110): five deeply-supervised losses,
``loss = (1 - epoch / n_epochs) * sum(side losses) + fused loss``, ``loss /= avg_grad_every_n``,
backward, step every ``avg_grad_every_n``-th iteration; snapshots every ``snapshot_every_n`` epochs.
"""
import os
import timeit
from pathlib import Path

import torch
from torch import optim

from config.mypath import Path as P
from layers.osvos_layers import class_balanced_cross_entropy_loss
from util import gpu_handler, io_helper, experiment_helper, args_helper
from util.logger import get_logger
from util.network_provider import NetworkProvider, provider_mapping
from util.settings import OfflineSettings
import parallel

log = get_logger(__file__)

save_dir_models = Path('models')
save_dir_results = Path('results')
db_root_dir = None
synthetic_size = None
data_parallel = False


def train_and_test(net_provider: NetworkProvider, settings: OfflineSettings) -> None:
    io_helper.write_settings(save_dir_models, net_provider.name, settings, variant_offline=settings.variant_offline)
    if settings.is_training:
        net_provider.load_network_train()
        shard = (parallel.rank(), parallel.world_size()) if data_parallel else None
        data_loader_train = io_helper.get_data_loader_train(db_root_dir, settings.batch_size_train,
                                                            synthetic=synthetic_size, shard=shard)
        data_loader_test = io_helper.get_data_loader_test(db_root_dir, settings.batch_size_test,
                                                          synthetic=synthetic_size)
        optimizer = net_provider.get_optimizer()
        summary_writer = _get_summary_writer()
        _train(net_provider, data_loader_train, data_loader_test, optimizer, summary_writer, settings.start_epoch,
               settings.n_epochs, settings.avg_grad_every_n, settings.snapshot_every_n,
               settings.is_testing_while_training, settings.test_every_n)

    if settings.is_testing:
        if not settings.is_training:
            net_provider.load_network_test()
        data_loader = io_helper.get_data_loader_test(db_root_dir, settings.batch_size_test, synthetic=synthetic_size)
        if settings.variant_offline is None:
            save_dir = save_dir_results / net_provider.name / 'offline'
        else:
            save_dir = save_dir_results / net_provider.name / str(settings.variant_offline) / 'offline'
        experiment_helper.test(net_provider, data_loader, save_dir, settings.is_visualizing_results,
                               settings.eval_speeds)


def _get_summary_writer():
    return io_helper.get_summary_writer(save_dir_models, comment='-offline')


def _losses(net, minibatch):
    inputs, gts = gpu_handler.cast_cuda_if_possible([minibatch['image'], minibatch['gt']])
    outputs = net.forward(inputs)
    # Data parallel: this minibatch is one rank's shard of the step's batch.  The reference balances the classes over the
    # whole batch tensor (src/layers/osvos_layers.py:28-39), so the two counts are summed over the ranks first (one
    # 16-byte all-reduce, shared by the five losses); the ranks' losses then add up to the single-process value.
    counts = parallel.batch_label_counts(gts) if data_parallel else None
    return [class_balanced_cross_entropy_loss(o, gts, size_average=False, batch_counts=counts) for o in outputs]


def _train(net_provider: NetworkProvider, data_loader_train, data_loader_test, optimizer: optim.SGD, summary_writer,
           start_epoch: int, n_epochs: int, avg_grad_every_n: int, snapshot_every_n: int,
           is_testing_while_training: bool, test_every_n: int) -> dict:
    log.info('Start of offline training')
    net = net_provider.network
    net.accumulate_grads_in_place = True  # this loop only ever calls loss.backward()
    # weights are constant inside an accumulation cycle: let the next forward overlap the wgrad tail of this backward
    net.defer_wgrad_join = os.environ.get('FOSVOS_DEFER_JOIN', '1') != '0'
    world = parallel.world_size() if data_parallel else 1
    # Data parallel here splits the BATCH, not the accumulation: every rank runs all
    # avg_grad_every_n iterations of a cycle on its own shard of each iteration's batch (global batch = world x
    # batch_size_train), the class counts of the loss are summed over the ranks per iteration (_losses) and the gradients
    # once per optimizer step - the update of a single process running the whole batch.
    local_accum = avg_grad_every_n
    # gradients live in one flat fp32 buffer: the wgrad kernels accumulate straight into it, zeroing is one memset,
    # and under data parallelism it is the single all-reduce payload
    named = list(net.named_parameters())
    flat = parallel.FlatGrads.attach(net, [p for _, p in named], names=[n for n, _ in named])
    sync = parallel.GradSync(net, flat)
    device = next(net.parameters()).device

    n_samples_train = len(data_loader_train)
    loss_train, loss_test, losses_train = [], [], []
    counter_gradient = 0
    n_iters = 0
    time_all_start = timeit.default_timer()
    for epoch in range(start_epoch, n_epochs):
        start_time = timeit.default_timer()
        # a DistributedSampler (data-parallel loader) draws the same permutation every epoch unless it is told the epoch;
        # the reference's shuffle=True loader draws a new order per epoch (src/util/io_helper.py:62-70)
        sampler = getattr(data_loader_train, 'sampler', None)
        if hasattr(sampler, 'set_epoch'):
            sampler.set_epoch(epoch)
        running = torch.zeros(5, device=device)
        for index, minibatch in enumerate(data_loader_train):
            losses = _losses(net, minibatch)
            running += torch.stack([l.detach() for l in losses])
            loss = (1 - epoch / n_epochs) * sum(losses[:-1]) + losses[-1]

            if index % n_samples_train == n_samples_train - 1:
                if world > 1:  # every rank holds its shards' part of the batch losses: the logged value is their sum
                    torch.distributed.all_reduce(running, op=torch.distributed.ReduceOp.SUM)
                vals = (running / n_samples_train).tolist()  # one device->host sync per epoch
                loss_train.append(vals[-1])
                losses_train.append(vals)  # all five deeply supervised losses of the epoch
                summary_writer.add_scalar('data/total_loss_epoch', vals[-1], epoch)
                log.info('[Epoch: %d, numImages: %5d]' % (epoch, index + 1))
                for l in range(len(vals)):
                    log.info('Loss %d: %f' % (l, vals[l]))
                log.info('Execution time: ' + str(timeit.default_timer() - start_time))

            # `loss /= nAveGrad; loss.backward()` of the reference, as a backward pass seeded with 1/nAveGrad (same gradient,
            # three fewer tiny kernels: see train_online._train)
            last_of_cycle = world > 1 and (counter_gradient + 1) % local_accum == 0
            if last_of_cycle:
                sync.arm()
            loss.backward(torch.full_like(loss.detach(), 1.0 / avg_grad_every_n))
            if last_of_cycle:
                sync.begin()
            counter_gradient += 1
            n_iters += 1

            if counter_gradient % local_accum == 0:
                net.join_gradients()
                sync.finish()  # the bucketed all-reduce begun right behind the cycle's last backward pass
                optimizer.step()
                flat.zero()
                counter_gradient = 0

        if (epoch % snapshot_every_n) == snapshot_every_n - 1 and epoch != 0 and parallel.rank() == 0:
            net_provider.save_model(epoch)

        if is_testing_while_training and epoch % test_every_n == (test_every_n - 1):
            with torch.no_grad():
                running_t = torch.zeros(5, device=device)
                for index, minibatch in enumerate(data_loader_test):
                    running_t += torch.stack(_losses(net, minibatch))
                vals = (running_t / max(len(data_loader_test), 1)).tolist()
            loss_test.append(vals[-1])
            summary_writer.add_scalar('data/test_loss_epoch', vals[-1], epoch)
            for l in range(len(vals)):
                log.info('***Testing *** Loss %d: %f' % (l, vals[l]))

    summary_writer.close()
    net.defer_wgrad_join = False  # joins
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return {'loss_train': loss_train, 'loss_test': loss_test, 'losses_train': losses_train, 'iterations': n_iters,
            'seconds': timeit.default_timer() - time_all_start}


def main(argv=None):
    global db_root_dir, synthetic_size, data_parallel
    args = args_helper.parse_args(is_online=False, argv=argv)
    if args.network != 'vgg16':
        raise SystemExit('only --network vgg16 is implemented on the HIP path (ResNet family: SURVEY.md §8 f4)')
    data_parallel = bool(args.data_parallel) and parallel.init_distributed()
    gpu_handler.select_gpu(args.gpu_id)
    db_root_dir = P.db_root_dir()
    synthetic_size = (args.height, args.width) if args.synthetic else None
    save_dir_models.mkdir(parents=True, exist_ok=True)
    save_dir_results.mkdir(parents=True, exist_ok=True)

    settings = OfflineSettings(is_training=args.is_training, is_testing=args.is_testing, start_epoch=0,
                               n_epochs=args.n_epochs or 240, avg_grad_every_n=args.avg_grad_every_n or 10,
                               snapshot_every_n=40, is_testing_while_training=False, test_every_n=5,
                               batch_size_train=1, batch_size_test=1, is_visualizing_network=False,
                               is_visualizing_results=False, is_loading_vgg_caffe=False,
                               variant_offline=args.variant_offline, eval_speeds=args.eval_speeds)
    provider_class = provider_mapping[('offline', args.network)]
    net_provider = provider_class(args.network, save_dir_models, settings, variant_offline=args.variant_offline)
    if args.synthetic:
        # no ImageNet weights offline (pretrained=1 needs torchvision + network): start from the reference's
        # random init instead and say so
        log.warning('--synthetic: starting from OSVOS_VGG(pretrained=0) random init')
        net_provider.load_network_train = lambda: net_provider.init_network(pretrained=0)
    train_and_test(net_provider, settings)


if __name__ == '__main__':
    main()
