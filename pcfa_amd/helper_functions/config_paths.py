"""Static configuration lookups with the reference's accessors -- `Paths.config(name)`, `Paths.splits(name)`,
`Conf.config(name)` (helper_functions/config_paths.py:1-35) -- filled from the environment instead of being
edited in place, so the same checkout runs from any working directory:

    PCFA_WEIGHTS_DIR   where the reference's checkpoint files live (default: models/_pretrained_weights under the cwd)
    PCFA_SINTEL_DIR / PCFA_KITTI15_DIR   dataset roots (readers are out of scope here, see datasets.py)
    PCFA_USE_CPU=1     Conf.config('useCPU') -> True
"""
import os

_ENV = os.environ.get


def _lookup(table, name):
    return table[name]


class Paths:
    @staticmethod
    def config(name):
        return _lookup({"sintel_mpi": _ENV("PCFA_SINTEL_DIR", ""), "kitti15": _ENV("PCFA_KITTI15_DIR", ""),
                        "weights": _ENV("PCFA_WEIGHTS_DIR", os.path.join("models", "_pretrained_weights"))}, name)

    @staticmethod
    def splits(name):
        return _lookup({"sintel_train": "training", "sintel_eval": "test", "kitti_train": "training",
                        "kitti_eval": "testing"}, name)


class Conf:
    @staticmethod
    def config(name):
        # correlationSamplerOnlyCPU exists for CLI/script compatibility only: the HIP cost-volume kernel runs where
        # the features live, the reference's CPU detour (config_paths.py:30) has no counterpart here.
        return _lookup({"useCPU": _ENV("PCFA_USE_CPU", "0") == "1", "correlationSamplerOnlyCPU": False}, name)
