"""Shared by the drop-in modules: the ctypes handle, pointer / stream helpers, status -> RuntimeError."""
import ctypes

import torch

from pcfa_amd import _hip


def lib():
    return _hip.load()


def P(t):
    return ctypes.c_void_p(t.data_ptr())


def S():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dense_gpu(*tensors):
    """The reference checks `is_cuda` and `is_contiguous` with TORCH_CHECK (correlation_sampler.cpp:31-33)."""
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("%s must be a CUDA tensor (libpcfa_hip.so has no CPU path)" % (tuple(t.shape),))
        if not t.is_contiguous():
            raise RuntimeError("tensor must be contiguous")
        if t.dtype != torch.float32:
            raise RuntimeError("tensor must be float32")


def check(status, prefix=""):
    if status:
        msg = lib().pcfa_status_string(status)
        raise RuntimeError(prefix + (msg.decode() if msg else "status %d" % status))
