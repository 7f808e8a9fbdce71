"""Model construction, checkpoints and the SGD recipe (reference: src/util/network_provider.py:18-159;
the ResNet providers at :162-528 are a different model family and out of scope).

Differences from the reference, all on purpose: ``save_dir`` is always the pair
(input model path, output directory); ``save_model`` writes a state_dict (what ``load_model`` reads);
``get_optimizer`` returns the fused HIP SGD, a ``torch.optim.SGD`` subclass with identical groups.
"""
from abc import ABC, abstractmethod
from pathlib import Path
from typing import Dict, Optional, Tuple, Type, Union

import torch
from torch import optim

from fosvos_hip.sgd import FusedSGD
from networks.osvos_vgg import OSVOS_VGG
from util import gpu_handler
from util.logger import get_logger
from util.settings import OfflineSettings, OnlineSettings, Settings

log = get_logger(__file__)


class NetworkProvider(ABC):
    def __init__(self, name: str, save_dir: Union[Path, Tuple[Path, Path]], network_type: type, settings: Settings,
                 variant_offline: Optional[int] = None, variant_online: Optional[int] = None) -> None:
        self.name = name
        if isinstance(save_dir, (tuple, list)):
            self.load_path, self.save_dir = Path(save_dir[0]), Path(save_dir[1])
        else:
            self.load_path, self.save_dir = None, Path(save_dir)
        self.network_type = network_type
        self._settings = settings
        self.variant_offline = variant_offline
        self.variant_online = variant_online
        self.network = None

    def init_network(self, **kwargs) -> object:
        net = self.network_type(**kwargs)
        net = gpu_handler.cast_cuda_if_possible(net, verbose=True)
        self.network = net
        return net

    def _get_file_path(self, epoch: int, sequence: Optional[str] = None) -> Path:
        model_name = self.name
        if self.variant_offline is not None:
            model_name += '_' + str(self.variant_offline)
        if sequence is not None:
            if self.variant_online is not None:
                model_name += '_' + str(self.variant_online)
            model_name += '_' + sequence
        return self.save_dir / '{0}_epoch-{1}.pth'.format(model_name, str(epoch))

    def load_model(self, epoch: int, sequence: Optional[str] = None) -> None:
        model_path = self.load_path if (self.load_path is not None and sequence is None) else self._get_file_path(epoch, sequence)
        log.info("Loading weights from: {0}".format(model_path))
        obj = torch.load(str(model_path), map_location=lambda storage, loc: storage)
        state = obj.state_dict() if isinstance(obj, torch.nn.Module) else obj  # accept both checkpoint kinds
        self.network.load_state_dict(state)
        self.network = gpu_handler.cast_cuda_if_possible(self.network, verbose=True)

    def save_model(self, epoch: int, sequence: Optional[str] = None) -> None:
        file_path = self._get_file_path(epoch, sequence)
        file_path.parent.mkdir(parents=True, exist_ok=True)
        log.info("Saving weights to: {0}".format(file_path))
        torch.save(self.network.state_dict(), str(file_path))

    @abstractmethod
    def load_network_train(self) -> None:
        pass

    @abstractmethod
    def load_network_test(self, sequence: Optional[str] = None) -> None:
        pass

    @abstractmethod
    def get_optimizer(self) -> optim.SGD:
        pass


def _named(mod, kind):
    return [p for n, p in mod.named_parameters() if kind in n]


class VGGOfflineProvider(NetworkProvider):
    def __init__(self, name: str, save_dir, settings: OfflineSettings, variant_offline: Optional[int] = None):
        super(VGGOfflineProvider, self).__init__(name=name, save_dir=save_dir, settings=settings,
                                                 network_type=OSVOS_VGG, variant_offline=variant_offline)

    def load_network_train(self) -> None:
        if self._settings.start_epoch == 0:
            self.init_network(pretrained=2 if self._settings.is_loading_vgg_caffe else 1)
        else:
            self.init_network(pretrained=0)
            self.load_model(self._settings.start_epoch)

    def load_network_test(self, sequence: Optional[str] = None) -> None:
        self.init_network(pretrained=0)
        self.load_model(self._settings.n_epochs, sequence=sequence)

    def get_optimizer(self, learning_rate: float = 1e-8, weight_decay: float = 0.0002,
                      momentum: float = 0.9) -> optim.SGD:
        net = self.network
        lr = learning_rate
        return FusedSGD([
            {'params': _named(net.stages, 'weight'), 'weight_decay': weight_decay, 'initial_lr': lr},
            {'params': _named(net.stages, 'bias'), 'lr': 2 * lr, 'initial_lr': 2 * lr},
            {'params': _named(net.side_prep, 'weight'), 'weight_decay': weight_decay, 'initial_lr': lr},
            {'params': _named(net.side_prep, 'bias'), 'lr': 2 * lr, 'initial_lr': 2 * lr},
            {'params': _named(net.score_dsn, 'weight'), 'lr': lr / 10, 'weight_decay': weight_decay,
             'initial_lr': lr / 10},
            {'params': _named(net.score_dsn, 'bias'), 'lr': 2 * lr / 10, 'initial_lr': 2 * lr / 10},
            {'params': _named(net.upscale, 'weight'), 'lr': 0, 'initial_lr': 0},
            {'params': _named(net.upscale_, 'weight'), 'lr': 0, 'initial_lr': 0},
            {'params': net.fuse.weight, 'lr': lr / 100, 'initial_lr': lr / 100, 'weight_decay': weight_decay},
            {'params': net.fuse.bias, 'lr': 2 * lr / 100, 'initial_lr': 2 * lr / 100},
        ], lr=lr, momentum=momentum)


class VGGOnlineProvider(NetworkProvider):
    def __init__(self, name: str, save_dir, settings: OnlineSettings, variant_offline: Optional[int] = None,
                 variant_online: Optional[int] = None):
        super(VGGOnlineProvider, self).__init__(name=name, save_dir=save_dir, settings=settings,
                                                network_type=OSVOS_VGG, variant_offline=variant_offline,
                                                variant_online=variant_online)

    def load_network_train(self) -> None:
        self.init_network(pretrained=0)
        self.load_model(self._settings.offline_epoch)

    def load_network_test(self, sequence: Optional[str] = None) -> None:
        self.init_network(pretrained=0)
        self.load_model(self._settings.n_epochs, sequence=sequence)

    def get_optimizer(self, learning_rate: float = 1e-8, weight_decay: float = 0.0002,
                      momentum: float = 0.9) -> optim.SGD:
        net = self.network
        lr = learning_rate
        return FusedSGD([
            {'params': _named(net.stages, 'weight'), 'weight_decay': weight_decay},
            {'params': _named(net.stages, 'bias'), 'lr': lr * 2},
            {'params': _named(net.side_prep, 'weight'), 'weight_decay': weight_decay},
            {'params': _named(net.side_prep, 'bias'), 'lr': lr * 2},
            {'params': _named(net.upscale, 'weight'), 'lr': 0},
            {'params': _named(net.upscale_, 'weight'), 'lr': 0},
            {'params': net.fuse.weight, 'lr': lr / 100, 'weight_decay': weight_decay},
            {'params': net.fuse.bias, 'lr': 2 * lr / 100},
        ], lr=lr, momentum=momentum)


provider_mapping = {
    ('offline', 'vgg16'): VGGOfflineProvider,
    ('online', 'vgg16'): VGGOnlineProvider,
}  # type: Dict[Tuple[str, str], Type[NetworkProvider]]
