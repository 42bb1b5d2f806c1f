"""Drop-in replacements for the reference's native extension modules, under the names the reference imports:

    spatial_correlation_sampler_backend   models/PWCNet/cpu_spatial_correlation_sampler-0.3.0/Correlation_Module/
                                          correlation_sampler.cpp:58-124 (pybind: forward / backward)
    correlation_cuda                      models/FlowNet/correlation_package/correlation_cuda.cc:10-167
    resample2d_cuda                       models/FlowNet/resample2d_package/resample2d_cuda.cc:6-24
    channelnorm_cuda                      models/FlowNet/channelnorm_package/channelnorm_cuda.cc:6-25

Put this directory on sys.path ahead of the built eggs (`install()` does it) and the reference's own Python sides
-- spatial_correlation_sampler.py:45-91, correlation.py:10-51, resample2d.py:12-43, channelnorm.py:11-36 -- run
unchanged on the MI355X: same positional signatures, same ownership (the sampler allocates its outputs like
at::zeros; the FlowNet ops fill / resize caller-allocated tensors and return 1), same error type (RuntimeError).
Every function launches on torch's current stream through the C-ABI of include/pcfa_hip.h; CPU tensors raise.
"""
import os
import sys


def install():
    """Make `import spatial_correlation_sampler_backend` (etc.) resolve to these modules."""
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    return here
