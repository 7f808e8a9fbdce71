"""Host plumbing around the loops (reference: src/util/io_helper.py): DataLoader factories, an
optional tensorboard writer and the YAML settings dump.  ``synthetic=(H, W)`` selects the synthetic sequence
instead of a DAVIS tree on disk."""
import dataclasses
from pathlib import Path
from typing import Optional, Tuple

import yaml
from torch.utils.data import DataLoader

from dataloaders import custom_transforms
from dataloaders.davis_2016 import DAVIS2016
from dataloaders.synthetic import SyntheticSequence
from util.logger import get_logger

log = get_logger(__file__)


class NullSummaryWriter:
    """Stands in for tensorboardX.SummaryWriter when it is not installed."""

    def add_scalar(self, *a, **k):
        pass

    def close(self):
        pass


def get_summary_writer(path, comment: str = ''):
    try:
        from tensorboardX import SummaryWriter
    except ImportError:
        return NullSummaryWriter()
    return SummaryWriter(log_dir=str(path), comment=comment)


def write_settings(save_dir: Path, name: str, settings, variant_offline: Optional[int] = None,
                   variant_online: Optional[int] = None) -> None:
    save_dir = Path(save_dir)
    save_dir.mkdir(parents=True, exist_ok=True)
    stem = name + ('' if variant_offline is None else '_' + str(variant_offline)) + \
        ('' if variant_online is None else '_' + str(variant_online))
    with open(str(save_dir / (stem + '_settings.yml')), 'w') as f:
        yaml.safe_dump(dataclasses.asdict(settings), f, default_flow_style=False)


def get_data_loader_train(db_root_dir, batch_size: int, seq_name: Optional[str] = None,
                          synthetic: Optional[Tuple[int, int]] = None,
                          shard: Optional[Tuple[int, int]] = None, resident: bool = True) -> DataLoader:
    """shard = (rank, world): this process draws its own 1/world of every epoch (data-parallel OFFLINE training, where
    the ranks split each iteration's batch); None: the reference's single-process loader.
    resident: a sequence run (``seq_name``: ONE training sample, src/dataloaders/davis_2016.py:72-83) gets the loader that
    keeps the sample's six flip / scale variants on the device and draws them with the reference pipeline's random numbers
    (dataloaders/resident.py: same tensors, same order as the DataLoader below under the same torch seed); False: that
    DataLoader itself (one worker start + decode + resample per iteration)."""
    if synthetic is not None:
        ds = SyntheticSequence(seq_name or 'synthetic', synthetic[0], synthetic[1], n_frames=1,
                               seed=1234 + (shard[0] if shard else 0))
        return DataLoader(ds, batch_size=batch_size, shuffle=True, num_workers=0)
    # src/util/io_helper.py:62-70: random flip, random rescale (ScaleNRotate stays disabled as in the reference), ToTensor
    composed = custom_transforms.Compose([custom_transforms.RandomHorizontalFlip(), custom_transforms.Resize(),
                                          custom_transforms.ToTensor()])
    if resident and seq_name is not None and shard is None and batch_size == 1:
        from dataloaders.resident import ResidentOneShotLoader
        return ResidentOneShotLoader(DAVIS2016(mode='train', db_root_dir=str(db_root_dir), transform=None, seq_name=seq_name))
    db_train = DAVIS2016(mode='train', db_root_dir=str(db_root_dir), transform=composed, seq_name=seq_name)
    if shard is not None:
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(db_train, num_replicas=shard[1], rank=shard[0], shuffle=True)
        return DataLoader(db_train, batch_size=batch_size, sampler=sampler, num_workers=1)
    return DataLoader(db_train, batch_size=batch_size, shuffle=True, num_workers=1)


def get_data_loader_test(db_root_dir, batch_size: int, seq_name: Optional[str] = None,
                         synthetic: Optional[Tuple[int, int]] = None, n_frames: int = 4) -> DataLoader:
    if synthetic is not None:
        ds = SyntheticSequence(seq_name or 'synthetic', synthetic[0], synthetic[1], n_frames=n_frames)
        return DataLoader(ds, batch_size=batch_size, shuffle=False, num_workers=0)
    db_test = DAVIS2016(mode='test', db_root_dir=str(db_root_dir), transform=custom_transforms.ToTensor(),
                        seq_name=seq_name)
    return DataLoader(db_test, batch_size=batch_size, shuffle=False, num_workers=2)
