"""FlowNet2's three native operators (models/FlowNet/{correlation,resample2d,channelnorm}_package/*.py)."""
import ctypes
import os
import weakref

import torch

from .. import _hip
from . import core
from .core import _call, _dev, _note_work, _pair, _ptr, _ptr_off, _stream


# --------------------------------------------------------------------------- #
# FlowNet2's native operators (models/FlowNet/{correlation,resample2d,channelnorm}_package)
# --------------------------------------------------------------------------- #
class CorrelationFunction(torch.autograd.Function):
    """correlation_package/correlation.py:10-51 on pcfa_flownet_corr_fwd/bwd (no rbot1/rbot2 scratch copies)."""

    @staticmethod
    def forward(ctx, input1, input2, pad_size=3, kernel_size=3, max_displacement=20, stride1=1, stride2=2,
                corr_multiply=1):
        _dev(input1, input2)
        input1, input2 = input1.contiguous(), input2.contiguous()
        if input1.shape != input2.shape or input1.dim() != 4:
            raise RuntimeError("Correlation: inputs must be two [B,C,H,W] tensors of the same shape")
        lib = _hip.load()
        B, C, H, W = input1.shape
        ctx.params = (B, C, H, W, int(pad_size), int(kernel_size), int(max_displacement), int(stride1), int(stride2))
        oc, oH, oW = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _hip.check(lib.pcfa_flownet_corr_out_size(H, W, *ctx.params[4:], ctypes.byref(oc), ctypes.byref(oH),
                                                  ctypes.byref(oW)), "pcfa_flownet_corr_out_size")
        out = torch.empty((B, oc.value, oH.value, oW.value), device=input1.device, dtype=torch.float32)
        _call("pcfa_flownet_corr_fwd", _ptr(input1), _ptr(input2), _ptr(out), *ctx.params)
        ctx.save_for_backward(input1, input2)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        g = grad_output.contiguous()
        g1, g2 = torch.empty_like(input1), torch.empty_like(input2)
        _call("pcfa_flownet_corr_bwd", _ptr(input1), _ptr(input2), _ptr(g), _ptr(g1), _ptr(g2), *ctx.params)
        return g1, g2, None, None, None, None, None, None


def flownet_correlation(input1, input2, pad_size=0, kernel_size=0, max_displacement=0, stride1=1, stride2=2,
                        corr_multiply=1):
    """Correlation.forward (correlation_package/correlation.py:53-67)."""
    return CorrelationFunction.apply(input1, input2, pad_size, kernel_size, max_displacement, stride1, stride2,
                                     corr_multiply)


class Resample2dFunction(torch.autograd.Function):
    """resample2d_package/resample2d.py:12-43."""

    @staticmethod
    def forward(ctx, input1, input2, kernel_size=1, bilinear=True):
        _dev(input1, input2)
        if not (input1.is_contiguous() and input2.is_contiguous()):
            raise AssertionError("Resample2d: inputs must be contiguous")  # the reference asserts (resample2d.py:16-17)
        B, C, iH, iW = input1.shape
        b, two, H, W = input2.shape
        if b != B or two != 2:
            raise RuntimeError("Resample2d: flow must be [B,2,H,W] with the batch size of input1")
        out = torch.empty((B, C, H, W), device=input1.device, dtype=torch.float32)
        ctx.params = (B, C, iH, iW, H, W, int(kernel_size), int(bool(bilinear)))
        _call("pcfa_resample2d_fwd", _ptr(input1), _ptr(input2), _ptr(out), *ctx.params)
        ctx.save_for_backward(input1, input2)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        g = grad_output.contiguous()
        g1, g2 = torch.empty_like(input1), torch.empty_like(input2)
        _call("pcfa_resample2d_bwd", _ptr(input1), _ptr(input2), _ptr(g), _ptr(g1), _ptr(g2), *ctx.params)
        return g1, g2, None, None


def resample2d(input1, input2, kernel_size=1, bilinear=True):
    """Resample2d.forward (resample2d_package/resample2d.py:45-56)."""
    return Resample2dFunction.apply(input1.contiguous(), input2, kernel_size, bilinear)


class ChannelNormFunction(torch.autograd.Function):
    """channelnorm_package/channelnorm.py:11-36."""

    @staticmethod
    def forward(ctx, input1, norm_deg=2):
        _dev(input1)
        if not input1.is_contiguous():
            raise AssertionError("ChannelNorm: input must be contiguous")  # channelnorm.py:15
        B, C, H, W = input1.shape
        out = torch.empty((B, 1, H, W), device=input1.device, dtype=torch.float32)
        ctx.params = (B, C, H * W, int(norm_deg))
        _call("pcfa_channelnorm_fwd", _ptr(input1), _ptr(out), *ctx.params)
        ctx.save_for_backward(input1, out)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        input1, out = ctx.saved_tensors
        g = grad_output.contiguous()
        g1 = torch.empty_like(input1)
        _call("pcfa_channelnorm_bwd", _ptr(input1), _ptr(out), _ptr(g), _ptr(g1), *ctx.params)
        return g1, None


def channelnorm(input1, norm_deg=2):
    """ChannelNorm.forward (channelnorm_package/channelnorm.py:38-45)."""
    return ChannelNormFunction.apply(input1, norm_deg)
