"""ORACLE (test infrastructure): build the REFERENCE's own C++ spatial correlation sampler.

Compiles, unmodified and where they lie, the two sources
    /root/reference/models/PWCNet/cpu_spatial_correlation_sampler-0.3.0/Correlation_Module/
        correlation.cpp  correlation_sampler.cpp
against the installed torch headers (the reference's own recipe is
`scripts/install_scs_cpu.sh` -> setup.py with CPU_ONLY=True; we call g++ through
torch.utils.cpp_extension instead of running its build system) into
    oracle/_ref/spatial_correlation_sampler_backend.so      (git-ignored, travels with gpurun)
Nothing is copied from the reference.  Used to pin oracle/spatial_corr.c (bit-exact check in
tests/test_oracle_cpu.py) and, optionally, as bench.py's "reference" CPU baseline for the
cost volume.  Skipped silently when /root/reference is absent (GPU box).
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/models/PWCNet/cpu_spatial_correlation_sampler-0.3.0/Correlation_Module"
OUT_DIR = os.path.join(HERE, "_ref")
NAME = "spatial_correlation_sampler_backend"


def so_path():
    return os.path.join(OUT_DIR, NAME + ".so")


def build(verbose=False):
    if os.path.exists(so_path()):
        return so_path()
    if not os.path.isdir(REF_SRC):
        return None
    from torch.utils.cpp_extension import load
    os.makedirs(OUT_DIR, exist_ok=True)
    build_dir = os.path.join(OUT_DIR, "_build")
    os.makedirs(build_dir, exist_ok=True)
    load(name=NAME, sources=[os.path.join(REF_SRC, "correlation.cpp"), os.path.join(REF_SRC, "correlation_sampler.cpp")],
         extra_cflags=["-fopenmp", "-O2"], extra_ldflags=["-lgomp"], build_directory=build_dir, verbose=verbose)
    shutil.copy(os.path.join(build_dir, NAME + ".so"), so_path())
    shutil.rmtree(build_dir, ignore_errors=True)
    return so_path()


def load_module():
    """Import the built extension (None if it does not exist)."""
    p = so_path()
    if not os.path.exists(p):
        return None
    import importlib.util
    import torch  # noqa: F401  (libtorch symbols must be loaded first)
    spec = importlib.util.spec_from_file_location(NAME, p)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules[NAME] = mod
    return mod


if __name__ == "__main__":
    print(build(verbose=True))
