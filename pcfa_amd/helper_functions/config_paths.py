"""Static configuration (reference helper_functions/config_paths.py:1-35), extended with
environment overrides so the same code runs from any working directory.

    PCFA_WEIGHTS_DIR   directory holding the reference's checkpoint files
                       (default: models/_pretrained_weights, relative to the cwd like the reference)
    PCFA_USE_CPU       "1" -> Conf.config('useCPU') is True
"""
import os


class Paths:
    __conf = {
        "sintel_mpi": os.environ.get("PCFA_SINTEL_DIR", ""),
        "kitti15": os.environ.get("PCFA_KITTI15_DIR", ""),
        "weights": os.environ.get("PCFA_WEIGHTS_DIR", os.path.join("models", "_pretrained_weights")),
    }
    __splits = {
        "sintel_train": "training",
        "sintel_eval": "test",
        "kitti_train": "training",
        "kitti_eval": "testing",
    }

    @staticmethod
    def config(name):
        return Paths.__conf[name]

    @staticmethod
    def splits(name):
        return Paths.__splits[name]


class Conf:
    __conf = {
        "useCPU": os.environ.get("PCFA_USE_CPU", "0") == "1",
        # The HIP cost-volume kernel runs where the features live; the reference's CPU detour
        # (correlationSamplerOnlyCPU: True) does not exist here.
        "correlationSamplerOnlyCPU": False,
    }

    @staticmethod
    def config(name):
        return Conf.__conf[name]
