"""Paths (reference: src/config/path_abstract.py + the user-written, git-ignored mypath.py).
Each location comes from an environment variable with a working default."""
import os


class Path:
    @staticmethod
    def db_root_dir():
        return os.environ.get('FOSVOS_DB_ROOT', './DAVIS')

    @staticmethod
    def save_root_dir():
        return os.environ.get('FOSVOS_SAVE_ROOT', './models')

    @staticmethod
    def exp_dir():
        return os.environ.get('FOSVOS_EXP_DIR', './')

    @staticmethod
    def models_dir():
        return os.environ.get('FOSVOS_MODELS_DIR', './models')

    @staticmethod
    def is_custom_pytorch():
        return False

    @staticmethod
    def custom_pytorch():
        return None

    @staticmethod
    def is_custom_opencv():
        return False

    @staticmethod
    def custom_opencv():
        return None
