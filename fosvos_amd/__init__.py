"""fosvos_amd: MI355X-native implementation of the OSVOS-VGG fine-tune path of klausondrag/FOSVOS.

The directory mirrors the reference's ``src/`` import root: put ``fosvos_amd/`` on ``sys.path`` and the
reference's own imports keep working (``from networks.osvos_vgg import OSVOS_VGG``,
``from layers.osvos_layers import class_balanced_cross_entropy_loss``, ``python train_online.py``).
Importing it as a package (``import fosvos_amd``) does that path setup for you.
"""
import os as _os
import sys as _sys

_ROOT = _os.path.dirname(_os.path.abspath(__file__))
if _ROOT not in _sys.path:
    _sys.path.insert(0, _ROOT)

__version__ = "0.1.0"
