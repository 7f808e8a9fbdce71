"""Per-module stream logger (reference: src/util/logger.py:9-23; colorlog is optional here)."""
import logging
from pathlib import Path

try:  # colours when colorlog is installed, plain logging otherwise
    import colorlog as _colorlog
except ImportError:
    _colorlog = None


def get_logger(file_name: str) -> logging.Logger:
    name = Path(file_name).stem
    logger = logging.getLogger('fosvos.' + name)
    if not logger.handlers:
        if _colorlog is not None:
            handler = _colorlog.StreamHandler()
            handler.setFormatter(_colorlog.ColoredFormatter('%(log_color)s%(levelname)s:%(name)s:%(message)s'))
        else:
            handler = logging.StreamHandler()
            handler.setFormatter(logging.Formatter('%(levelname)s:%(name)s:%(message)s'))
        logger.addHandler(handler)
        logger.setLevel(logging.WARNING)
        logger.propagate = False
    return logger
