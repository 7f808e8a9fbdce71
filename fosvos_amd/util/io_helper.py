"""Host plumbing around the loops (reference: src/util/io_helper.py): DataLoader factories, an
optional tensorboard writer and the YAML settings dump.  DAVIS itself is a "next" row;
``synthetic=(H, W)`` selects the synthetic sequence."""
import dataclasses
from pathlib import Path
from typing import Optional, Tuple

import yaml
from torch.utils.data import DataLoader

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


def _davis(train: bool, db_root_dir, seq_name):
    try:
        from dataloaders.davis_2016 import DAVIS2016  # not built yet
    except ImportError as e:
        raise RuntimeError("the DAVIS2016 loader is not part of this build yet; run with --synthetic") from e
    return DAVIS2016(train=train, db_root_dir=db_root_dir, seq_name=seq_name)


def get_data_loader_train(db_root_dir, batch_size: int, seq_name: Optional[str] = None,
                          synthetic: Optional[Tuple[int, int]] = None) -> DataLoader:
    if synthetic is not None:
        ds = SyntheticSequence(seq_name or 'synthetic', synthetic[0], synthetic[1], n_frames=1)
        return DataLoader(ds, batch_size=batch_size, shuffle=True, num_workers=0)
    return DataLoader(_davis(True, db_root_dir, seq_name), batch_size=batch_size, shuffle=True, num_workers=1)


def get_data_loader_test(db_root_dir, batch_size: int, seq_name: Optional[str] = None,
                         synthetic: Optional[Tuple[int, int]] = None, n_frames: int = 4) -> DataLoader:
    if synthetic is not None:
        ds = SyntheticSequence(seq_name or 'synthetic', synthetic[0], synthetic[1], n_frames=n_frames)
        return DataLoader(ds, batch_size=batch_size, shuffle=False, num_workers=0)
    return DataLoader(_davis(False, db_root_dir, seq_name), batch_size=batch_size, shuffle=False, num_workers=2)
