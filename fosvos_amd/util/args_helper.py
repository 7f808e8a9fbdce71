"""Command-line flags of the train scripts (reference: src/util/args_helper.py:5-39).

Same flag names and meanings.  The reference declares ``type=Optional[str]`` for -s/-sg/-sgs, which
argparse cannot call; the intended types are used here.  Extensions (not in the
reference, all optional): --n-epochs, --avg-grad-every-n, --synthetic, --height/--width, --parent-model,
--data-parallel.
"""
import argparse
from typing import List, Optional


def _get_base_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(add_help=True)
    parser.add_argument('--gpu-id', default=None, type=int, help='The gpu id to use')
    parser.add_argument('--network', default='vgg16', type=str, choices=['vgg16', 'resnet18', 'resnet34'],
                        help='The network to use (only vgg16 is implemented on the HIP path)')
    parser.add_argument('--no-training', action='store_true', help='skip training')
    parser.add_argument('--no-testing', action='store_true', help='skip testing')
    parser.add_argument('--variant-offline', default=None, type=int, help='version to try')
    parser.add_argument('--eval-speeds', action='store_true', help='evaluates the network speeds')
    # ---- extensions
    parser.add_argument('--n-epochs', default=None, type=int, help='override the hard-coded epoch count')
    parser.add_argument('--avg-grad-every-n', default=None, type=int, help='override gradient accumulation length')
    parser.add_argument('--synthetic', action='store_true',
                        help='run on synthetic frames of --height x --width instead of DAVIS')
    parser.add_argument('--height', default=480, type=int)
    parser.add_argument('--width', default=854, type=int)
    parser.add_argument('--parent-model', default=None, type=str, help='state_dict (.pth) of the parent network')
    parser.add_argument('--data-parallel', action='store_true',
                        help='spread the gradient-accumulation micro-batches over the ranks of a torchrun job '
                             '(RCCL all-reduce per optimizer step)')
    return parser


def parse_args(is_online: bool, argv: Optional[List[str]] = None) -> argparse.Namespace:
    parser = _get_base_parser()
    if is_online:
        parser.add_argument('-s', '--sequence-name', default=None, type=str)
        parser.add_argument('-sg', '--sequence-group', default=None, type=int)
        parser.add_argument('-sgs', '--sequence-group-size', default=None, type=int)
        parser.add_argument('--variant-online', default=None, type=int, help='version to try')
    args = parser.parse_args(argv)
    args.is_training = not args.no_training
    args.is_testing = not args.no_testing
    return args
