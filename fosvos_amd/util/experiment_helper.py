"""Inference pass over a sequence (reference: src/util/experiment_helper.py:20-80): forward, sigmoid,
write probability PNGs; with ``eval_speeds`` time ``net.forward`` between device synchronisations
over 10 passes, dropping the first minibatch of each pass (the reference's protocol, :29-53,77-80;
no PNGs are written in that mode, as in the reference)."""
import timeit
from pathlib import Path
from typing import Optional

import numpy as np
import torch
from torch import cuda

from util import gpu_handler
from util.logger import get_logger

log = get_logger(__file__)

# what the last call of test() did: {'n_runs', 'n_forward', 'times' (seconds, the kept samples), 'accurate_images',
# 'time_per_sample'}.  The reference only logs these numbers (:70-80); tests and bench.py read them here.
last_eval = {}


def bytescale(data: np.ndarray) -> np.ndarray:
    """What ``scipy.misc.imsave`` did to a float image before writing it (reference: src/util/experiment_helper.py:64
    calls it on the sigmoid map; scipy 1.0/1.1 ``misc.pilutil``: imsave -> toimage -> bytescale with cmin = data.min(),
    cmax = data.max(), low = 0, high = 255): the map is stretched to ITS OWN value range, then rounded half up."""
    data = np.asarray(data, dtype=np.float64)
    cmin, cmax = float(data.min()), float(data.max())
    cscale = cmax - cmin
    if cscale == 0:
        cscale = 1.0
    scaled = (data - cmin) * (255.0 / cscale)
    return (scaled.clip(0, 255) + 0.5).astype(np.uint8)


def _save_png(path: Path, prob: np.ndarray) -> None:
    from PIL import Image
    Image.fromarray(bytescale(prob), mode='L').save(str(path))


def test(net_provider, data_loader, save_dir: Path, is_visualizing_results: bool, eval_speeds: bool,
         seq_name: Optional[str] = None):
    log.info('Testing Network')
    net = net_provider.network
    n_runs = 10 if eval_speeds else 1
    times = []
    n_forward = 0
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
                n_forward += 1
                if eval_speeds:
                    cuda.synchronize()
                    if minibatch_index > 0:  # first allocate takes longer
                        times.append(timeit.default_timer() - time_image_start)
                else:
                    # reference :57-59: 1 / (1 + exp(-pred)) in numpy on the host
                    pred = outputs[-1].cpu().numpy()
                    probs = 1.0 / (1.0 + np.exp(-pred))
                    for index in range(inputs.size()[0]):
                        save_dir_seq = Path(save_dir) / minibatch_seq_name[index]
                        save_dir_seq.mkdir(parents=True, exist_ok=True)
                        _save_png(save_dir_seq / '{0}.png'.format(fname[index]), probs[index, 0])
    time_for_all = timeit.default_timer() - time_all_start
    n_images = len(data_loader)
    time_per_sample = time_for_all / max(n_images, 1)
    log.info('Test {0}: total test time {1} sec'.format(seq_name, str(time_for_all)))
    log.info('Test {0}: {1} images'.format(seq_name, str(n_images)))
    log.info('Test {0}: time per sample {1} sec'.format(seq_name, str(time_per_sample)))
    last_eval.clear()
    last_eval.update(n_runs=n_runs, n_forward=n_forward, times=list(times), accurate_images=(n_images - 1) * n_runs,
                     time_per_sample=time_per_sample)
    if eval_speeds and times:
        log.info('Test {0}: accurate {1} images'.format(seq_name, str((n_images - 1) * n_runs)))
        log.info('Test {0}: accurate total time {1} sec ({2} runs)'.format(seq_name, np.sum(times), n_runs))
        log.info('Test {0}: accurate time per sample {1} sec ({2} runs)'.format(seq_name, np.average(times), n_runs))
        return float(np.average(times))
    return None
