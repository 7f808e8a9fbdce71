"""The one-shot training sample, resident on the device.

In a sequence run the reference's training loader holds exactly ONE sample - frame 00000 and its mask
(src/dataloaders/davis_2016.py:72-83) - behind ``DataLoader(db_train, batch_size=1, shuffle=True, num_workers=1)``
(src/util/io_helper.py:62-70): every iteration starts a worker process, decodes the JPEG and the PNG, draws a flip
(src/dataloaders/custom_transforms.py:96-111) and one of three scales (:63-93), resamples, and copies the result to the
device: several milliseconds per iteration in front of a 0.8 ms training step.  The augmentation has only
2 flips x 3 scales = SIX outcomes, all determined by that one sample.

``ResidentOneShotLoader`` decodes the sample once, builds the six variants once with the SAME transform code, keeps them
on the device, and per epoch draws the variant with the reference pipeline's random numbers in the reference pipeline's
order, so that it yields, tensor for tensor, what the per-iteration DataLoader yields under the same torch seed:

  * creating the loader's iterator draws the workers' base seed from torch's default generator
    (``torch.empty((), dtype=torch.int64).random_()``);
  * worker 0 seeds Python's ``random`` with ``base_seed + 0``; fetching the sample then calls
    ``random.random()`` (flip if < 0.5) and ``random.randint(0, 2)`` (index into the scales), in that order;
  * the shuffling sampler draws one more int64 from the default generator when its first index is asked for.
"""
import random
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from dataloaders import custom_transforms


class ResidentOneShotLoader(object):
    """Drop-in for the training DataLoader of a one-sample dataset: ``len() == 1``, each ``iter()`` yields one minibatch
    dict (``image`` [1,3,h,w], ``gt`` [1,1,h,w] on ``device``, ``seq_name`` / ``fname`` lists as default_collate builds them)."""

    def __init__(self, dataset, device: Optional[torch.device] = None, scales: Sequence[float] = (0.5, 0.8, 1)):
        if len(dataset) != 1:
            raise ValueError("ResidentOneShotLoader holds the single sample of a one-shot sequence run, got %d samples"
                             % len(dataset))
        if dataset.transform is not None:
            raise ValueError("pass the dataset without its transform: the loader applies flip / rescale / ToTensor itself")
        self.dataset = dataset
        self.scales = list(scales)
        self.device = device if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        base = dataset[0]  # decoded once
        self._meta = {k: [base[k]] for k in ("seq_name", "fname") if k in base}
        to_tensor = custom_transforms.ToTensor()
        self.variants: Dict[tuple, Dict[str, torch.Tensor]] = {}
        for flip in (False, True):
            for si, sc in enumerate(self.scales):
                sample = {"image": base["image"], "gt": base["gt"]}
                if flip:  # custom_transforms.RandomHorizontalFlip with the draw fixed
                    sample = {k: np.ascontiguousarray(v[:, ::-1]) for k, v in sample.items()}
                sample = {k: custom_transforms.resize(v, sc, sc) for k, v in sample.items()}  # custom_transforms.Resize
                sample = to_tensor(sample)
                self.variants[(flip, si)] = {k: v.unsqueeze(0).to(self.device) for k, v in sample.items()}
        self.draws: List[tuple] = []  # (flip, scale index) of every epoch so far

    def __len__(self) -> int:
        return 1

    def __iter__(self):
        # _BaseDataLoaderIter.__init__: the workers' base seed, from torch's default generator
        base_seed = int(torch.empty((), dtype=torch.int64).random_().item())
        rng = random.Random(base_seed + 0)        # worker 0: random.seed(base_seed + worker_id)
        flip = rng.random() < 0.5                 # RandomHorizontalFlip.__call__
        si = rng.randint(0, len(self.scales) - 1)  # Resize.__call__
        torch.empty((), dtype=torch.int64).random_()  # RandomSampler.__iter__: its own seed, drawn at the first index
        self.draws.append((flip, si))
        batch = dict(self.variants[(flip, si)])
        batch.update(self._meta)
        yield batch
