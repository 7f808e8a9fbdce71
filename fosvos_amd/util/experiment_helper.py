"""Inference pass over a sequence (reference: src/util/experiment_helper.py:20-80): forward, sigmoid,
write probability PNGs; with ``eval_speeds`` time ``net.forward`` between device synchronisations
over 10 passes, dropping the first minibatch of each pass (the reference's protocol, :29-53,77-80)."""
import timeit
from pathlib import Path
from typing import Optional

import numpy as np
import torch
from torch import cuda

from util import gpu_handler
from util.logger import get_logger

log = get_logger(__file__)


def _save_png(path: Path, prob: np.ndarray) -> None:
    from PIL import Image
    Image.fromarray(np.clip(prob * 255.0 + 0.5, 0, 255).astype(np.uint8)).save(str(path))


def test(net_provider, data_loader, save_dir: Path, is_visualizing_results: bool, eval_speeds: bool,
         seq_name: Optional[str] = None):
    log.info('Testing Network')
    net = net_provider.network
    n_runs = 10 if eval_speeds else 1
    times = []
    time_all_start = timeit.default_timer()
    with torch.no_grad():
        for _ in range(n_runs):
            for minibatch_index, minibatch in enumerate(data_loader):
                img, gt = minibatch['image'], minibatch['gt']
                minibatch_seq_name, fname = minibatch['seq_name'], minibatch['fname']
                inputs, gts = gpu_handler.cast_cuda_if_possible([img, gt])
                if eval_speeds:
                    cuda.synchronize()
                    time_image_start = timeit.default_timer()
                outputs = net.forward(inputs)
                if eval_speeds:
                    cuda.synchronize()
                    if minibatch_index > 0:  # first allocate takes longer
                        times.append(timeit.default_timer() - time_image_start)
                else:
                    probs = torch.sigmoid(outputs[-1]).cpu().numpy()
                    for index in range(inputs.size()[0]):
                        save_dir_seq = Path(save_dir) / minibatch_seq_name[index]
                        save_dir_seq.mkdir(parents=True, exist_ok=True)
                        _save_png(save_dir_seq / '{0}.png'.format(fname[index]), probs[index, 0])
    time_for_all = timeit.default_timer() - time_all_start
    n_images = len(data_loader)
    log.info('Test {0}: total test time {1} sec'.format(seq_name, str(time_for_all)))
    log.info('Test {0}: {1} images'.format(seq_name, str(n_images)))
    log.info('Test {0}: time per sample {1} sec'.format(seq_name, str(time_for_all / max(n_images, 1))))
    if eval_speeds and times:
        log.info('Test {0}: accurate time per sample {1} sec ({2} runs)'.format(seq_name, np.average(times), n_runs))
        return float(np.average(times))
    return None
