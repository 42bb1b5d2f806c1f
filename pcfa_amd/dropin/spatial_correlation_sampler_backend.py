"""`spatial_correlation_sampler_backend` on libpcfa_hip.so: the pybind module of the reference's sampler
(Correlation_Module/correlation_sampler.cpp:58-81 forward, :83-112 backward, :114-124 bindings), same 12 integers
in the same order."""
import ctypes

import torch

from pcfa_amd.dropin._common import P, S, check, dense_gpu, lib


def forward(input1, input2, kH, kW, patchH, patchW, padH, padW, dilationH, dilationW, dilation_patchH,
            dilation_patchW, dH, dW):
    dense_gpu(input1, input2)
    B, C, iH, iW = input1.shape
    oH, oW = ctypes.c_int(), ctypes.c_int()
    check(lib().pcfa_spatial_corr_out_size(iH, iW, kH, kW, padH, padW, dilationH, dilationW, dH, dW,
                                           ctypes.byref(oH), ctypes.byref(oW)))
    out = input1.new_empty(B, patchH, patchW, oH.value, oW.value)   # callee-allocated, like at::zeros (:95)
    check(lib().pcfa_spatial_corr_fwd(P(input1), P(input2), P(out), B, C, iH, iW, kH, kW, patchH, patchW, padH, padW,
                                      dilationH, dilationW, dilation_patchH, dilation_patchW, dH, dW, S()))
    return out


def backward(input1, input2, grad_output, kH, kW, patchH, patchW, padH, padW, dilationH, dilationW,
             dilation_patchH, dilation_patchW, dH, dW):
    dense_gpu(input1, input2)
    grad_output = grad_output.contiguous()
    g1, g2 = torch.empty_like(input1), torch.empty_like(input2)
    B, C, iH, iW = input1.shape
    check(lib().pcfa_spatial_corr_bwd(P(input1), P(input2), P(grad_output), P(g1), P(g2), B, C, iH, iW, kH, kW,
                                      patchH, patchW, padH, padW, dilationH, dilationW, dilation_patchH,
                                      dilation_patchW, dH, dW, S()))
    return [g1, g2]
