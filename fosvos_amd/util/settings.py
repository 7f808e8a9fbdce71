"""Run settings (reference: src/util/settings.py:4-30; same field names, plain dataclasses)."""
from dataclasses import dataclass
from typing import Optional


@dataclass
class Settings:
    is_training: bool
    is_testing: bool
    start_epoch: int
    n_epochs: int
    avg_grad_every_n: int
    snapshot_every_n: int
    is_testing_while_training: bool
    test_every_n: int
    batch_size_train: int
    batch_size_test: int
    is_visualizing_network: bool
    is_visualizing_results: bool
    variant_offline: Optional[int]
    eval_speeds: bool


@dataclass
class OfflineSettings(Settings):
    is_loading_vgg_caffe: bool = False


@dataclass
class OnlineSettings(Settings):
    offline_epoch: int = 240
    variant_online: Optional[int] = None
