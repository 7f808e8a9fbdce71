"""Device selection (reference: src/util/gpu_handler.py).  torch.cuda IS HIP on ROCm, so the calls
are unchanged; the reference's author-hostname table is replaced by LOCAL_RANK / device 0."""
import os
from typing import List, Optional, Union

import torch
from torch.nn import Module

from util.logger import get_logger

log = get_logger(__file__)

_gpu_id_default_value = 0


def select_gpu_by_id(gpu_id: int = _gpu_id_default_value) -> None:
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: the HIP path has no CPU fallback")
    log.info('Using GPU {} {}'.format(str(gpu_id), torch.cuda.get_device_name(gpu_id)))
    torch.cuda.set_device(device=gpu_id)


def select_gpu_by_hostname(hostname: Optional[str] = None) -> None:
    select_gpu_by_id(int(os.environ.get('LOCAL_RANK', _gpu_id_default_value)))


def select_gpu(gpu_id: Optional[int] = None) -> None:
    if gpu_id is None:
        select_gpu_by_hostname()
    else:
        select_gpu_by_id(gpu_id)


def cast_cuda_if_possible(net: Union[Module, List[Module], torch.Tensor, List[torch.Tensor]], verbose: bool = False):
    if torch.cuda.is_available():
        if verbose:
            log.info('Using cuda')
        if type(net) is list:
            return [n.cuda() for n in net]
        return net.cuda()
    if verbose:
        log.warning('Not using cuda')
    return net
