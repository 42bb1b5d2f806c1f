"""PWC-Net operators: the 9x9 cost volume through the spatial-correlation sampler boundary
(spatial_correlation_sampler.py:9-91, PWCNet.py:45-58), the backward warp (PWCNet.py:166-206), the deconv / upfeat
layers (PWCNet.py:42-43) and the final bilinear up-sampling (PWCNet.py:73,321)."""
import ctypes
import os
import weakref

import torch

from .. import _hip
from . import core
from .core import _call, _dev, _note_work, _pair, _ptr, _ptr_off, _stream


# --------------------------------------------------------------------------- #
# PWC-Net cost volume
# --------------------------------------------------------------------------- #
class SpatialCorrelationSamplerFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input1, input2, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1,
                dilation_patch=1):
        _dev(input1, input2)
        # the reference's CPU build reads through accessors (any strides); the kernels want dense NCHW
        input1, input2 = input1.contiguous(), input2.contiguous()
        lib = _hip.load()
        kH, kW = _pair(kernel_size)
        pH, pW = _pair(patch_size)
        padH, padW = _pair(padding)
        dilH, dilW = _pair(dilation)
        dpH, dpW = _pair(dilation_patch)
        dH, dW = _pair(stride)
        B, C, iH, iW = input1.shape
        oH, oW = ctypes.c_int(), ctypes.c_int()
        _hip.check(lib.pcfa_spatial_corr_out_size(iH, iW, kH, kW, padH, padW, dilH, dilW, dH, dW,
                                                  ctypes.byref(oH), ctypes.byref(oW)), "pcfa_spatial_corr_out_size")
        out = torch.empty((B, pH, pW, oH.value, oW.value), device=input1.device, dtype=torch.float32)
        ctx.params = (B, C, iH, iW, kH, kW, pH, pW, padH, padW, dilH, dilW, dpH, dpW, dH, dW)
        _call("pcfa_spatial_corr_fwd", _ptr(input1), _ptr(input2), _ptr(out), *ctx.params)
        ctx.save_for_backward(input1, input2)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        lib = _hip.load()
        g = grad_output.contiguous()
        g1 = torch.empty_like(input1)
        g2 = torch.empty_like(input2)
        _call("pcfa_spatial_corr_bwd", _ptr(input1), _ptr(input2), _ptr(g), _ptr(g1), _ptr(g2),
                                             *ctx.params)
        return g1, g2, None, None, None, None, None, None


def spatial_correlation_sample(input1, input2, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1,
                               dilation_patch=1):
    return SpatialCorrelationSamplerFunction.apply(input1, input2, kernel_size, patch_size, stride, padding,
                                                   dilation, dilation_patch)


class _PwcCostVolume(torch.autograd.Function):
    """leaky_relu(spatial_correlation_sample(a, b, patch 9) / C) -- PWCNet.py:45-58 + the LeakyReLU that follows every
    call (:249,264,278,292,308) -- as ONE forward launch (scale and activation in the epilogue) and ONE backward
    launch (mask * scale applied to the gradient taps while they are staged; both input gradients)."""

    @staticmethod
    def forward(ctx, input1, input2, slope):
        _dev(input1, input2)
        input1, input2 = input1.contiguous(), input2.contiguous()
        B, C, H, W = input1.shape
        out = torch.empty((B, 81, H, W), device=input1.device, dtype=torch.float32)
        ctx.args = (B, C, H, W, 1.0 / C, float(slope))
        _call("pcfa_cost_volume9_fwd", _ptr(input1), _ptr(input2), _ptr(out), *ctx.args)
        ctx.save_for_backward(input1, input2, out)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        input1, input2, out = ctx.saved_tensors
        g = grad_output.contiguous()
        if g.data_ptr() % 16:          # a view with a storage offset: the kernel stages 16-B pieces
            g = g.clone()
        g1, g2 = torch.empty_like(input1), torch.empty_like(input2)
        _call("pcfa_cost_volume9_bwd", _ptr(input1), _ptr(input2), _ptr(out), _ptr(g), _ptr(g1), _ptr(g2), *ctx.args)
        return g1, g2, None


def pwc_cost_volume(input1, input2, slope=0.1):
    """PWC-Net's `leakyRELU(correlate(input1, input2))`: [B,C,H,W] x2 -> [B,81,H,W]."""
    misaligned = any(t.is_contiguous() and t.data_ptr() % 16 for t in (input1, input2))
    if input1.shape[-1] % 4 != 0 or input1.shape != input2.shape or misaligned:   # the fused kernels stage 16-B pieces
        out = spatial_correlation_sample(input1, input2, kernel_size=1, patch_size=9, stride=1)
        b, ph, pw, h, w = out.size()
        return torch.nn.functional.leaky_relu(out.view(b, ph * pw, h, w) / input1.size(1), slope)
    return _PwcCostVolume.apply(input1, input2, slope)


class _PwcWarp(torch.autograd.Function):
    """PWCDCNet.warp (models/PWCNet/PWCNet.py:166-206) on pcfa_pwc_warp_fwd/bwd."""

    @staticmethod
    def forward(ctx, x, flo, mask_threshold, deterministic=True, flow_scale=1.0):
        _dev(x, flo)
        ctx.deterministic = bool(deterministic)
        x, flo = x.contiguous(), flo.contiguous()
        B, C, H, W = x.shape
        if tuple(flo.shape) != (B, 2, H, W):
            raise ValueError("pwc_warp: flow %s does not match features %s" % (tuple(flo.shape), tuple(x.shape)))
        out = torch.empty_like(x)
        ctx.params = (B, C, H, W, float(mask_threshold), float(flow_scale))
        _call("pcfa_pwc_warp_fwd", _ptr(x), _ptr(flo), _ptr(out), *ctx.params)
        ctx.save_for_backward(x, flo)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        x, flo = ctx.saved_tensors
        g = g.contiguous()
        gx, gf = torch.empty_like(x), torch.empty_like(flo)
        if ctx.deterministic:   # fixed-point scatter (bit-reproducible); else hardware fp32 atomics
            B, C, H, W, thr, fs = ctx.params
            nws = int(_hip.load().pcfa_pwc_warp_bwd_det_workspace_bytes(B, C, H, W))
            ws = torch.empty((nws + 7) // 8, device=x.device, dtype=torch.int64)
            _call("pcfa_pwc_warp_bwd_det", _ptr(x), _ptr(flo), _ptr(g), _ptr(gx), _ptr(gf), _ptr(ws), nws, B, C, H, W,
                  thr, fs)
        else:
            _call("pcfa_pwc_warp_bwd", _ptr(x), _ptr(flo), _ptr(g), _ptr(gx), _ptr(gf), *ctx.params)
        return gx, gf, None, None, None


def pwc_warp(x, flo, mask_threshold=0.0001, deterministic=True, flow_scale=1.0):
    """Backward-warp x by flow_scale * flo with PWC-Net's validity mask: one launch forward, three backward.  deterministic
    (Config.warp_bwd_deterministic): the backward scatters fixed-point values (bit-reproducible); False: fp32 atomics.
    flow_scale: `self.warp(c2, up_flow * 0.625)` (PWCNet.py:262,276,290,306) without the two element-wise launches; the
    same bits (the product is rounded before use, the flow gradient is multiplied after the sum)."""
    return _PwcWarp.apply(x, flo, mask_threshold, deterministic, flow_scale)


class _SplitBatch(torch.autograd.Function):
    """(x[:b], x[b:]) as views; the backward writes both gradients into one tensor with ONE concatenation (autograd's own
    slice backward is a fill + copy per half and an add)."""

    @staticmethod
    def forward(ctx, x, b):
        ctx.dims = (b, tuple(x.shape))
        ctx.set_materialize_grads(False)
        return x[:b], x[b:]

    @staticmethod
    def backward(ctx, g0, g1):
        b, shape = ctx.dims
        if g0 is None and g1 is None:
            return None, None
        ref = g0 if g0 is not None else g1
        if g0 is None:
            g0 = ref.new_zeros((b,) + shape[1:])
        if g1 is None:
            g1 = ref.new_zeros((shape[0] - b,) + shape[1:])
        return torch.cat((g0, g1), 0), None


def split_batch(x, b):
    """x[:b], x[b:] -- the two images' halves of a feature tensor computed for both at once (nets/pwcnet.py runs the
    feature pyramid of PWCNet.py:233-244 on cat((im1, im2), 0): half the launches, twice the workgroups per launch)."""
    return _SplitBatch.apply(x, b)


class _Deconv4s2FewOut(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        _dev(x, weight, bias)
        if weight.dim() != 4 or tuple(weight.shape[2:]) != (4, 4) or not 1 <= weight.shape[1] <= 4:
            raise ValueError("deconv4s2_fewout expects a [K, N<=4, 4, 4] ConvTranspose2d weight, got %s"
                             % (tuple(weight.shape),))
        x = x.contiguous()
        B, K, H, W = x.shape
        N = weight.shape[1]
        if weight.shape[0] != K:
            raise ValueError("deconv4s2_fewout: input %s does not match weight %s" % (tuple(x.shape), tuple(weight.shape)))
        w = weight.detach().contiguous()
        out = torch.empty((B, N, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
        nws = int(_hip.load().pcfa_deconv4s2_fewout_workspace_bytes(B, K, N, H, W))
        ws = torch.empty(nws // 4, device=x.device, dtype=torch.float32) if nws else None
        _call("pcfa_deconv4s2_fewout_fwd", _ptr(x), _ptr(w), _ptr(bias), _ptr(out), _ptr(ws), B, K, N, H, W)
        ctx.save_for_backward(w)
        ctx.dims = (B, K, N, H, W)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            raise RuntimeError("deconv4s2_fewout is the frozen-weight path: no weight / bias gradient")
        (w,) = ctx.saved_tensors
        B, K, N, H, W = ctx.dims
        g = g.contiguous()
        gx = torch.empty((B, K, H, W), device=g.device, dtype=torch.float32)
        _call("pcfa_deconv4s2_fewout_bwd", _ptr(g), _ptr(w), _ptr(gx), B, K, N, H, W)
        return gx, None, None


def deconv4s2_fewout(x, weight, bias=None):
    """conv_transpose2d(x, weight, bias, stride=2, padding=1) for a frozen 4x4 weight with at most 4 output channels:
    PWC-Net's deconv / upfeat layers (PWCNet.py:42-43) as a streaming kernel with a fixed summation order."""
    return _Deconv4s2FewOut.apply(x, weight, bias)


class _UpsampleBilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, factor, mul):
        _dev(x)
        x = x.contiguous()
        B, C, H, W = x.shape
        out = torch.empty((B, C, factor * H, factor * W), device=x.device, dtype=torch.float32)
        _call("pcfa_upsample_bilinear_fwd", _ptr(x), _ptr(out), B * C, H, W, int(factor), float(mul))
        ctx.dims = (B, C, H, W, int(factor), float(mul))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        B, C, H, W, factor, mul = ctx.dims
        g = g.contiguous()
        gx = torch.empty((B, C, H, W), device=g.device, dtype=torch.float32)
        _call("pcfa_upsample_bilinear_bwd", _ptr(g), _ptr(gx), B * C, H, W, factor, mul)
        return gx, None, None


def upsample_bilinear(x, factor, mul=1.0):
    """mul * nn.Upsample(scale_factor=factor, mode='bilinear')(x) (PWCNet.py:73,321); gather backward (no atomics)."""
    return _UpsampleBilinear.apply(x, int(factor), float(mul))
