"""`correlation_cuda` on libpcfa_hip.so (models/FlowNet/correlation_package/correlation_cuda.cc:10-87 forward,
:89-167 backward): caller-allocated outputs that the callee resizes, returns 1, launch failure -> RuntimeError with
the reference's "CUDA call failed" text.  rbot1 / rbot2 (the reference's zero-padded channels-last scratch copies)
are accepted and left untouched: the kernels pad by predicate."""
import ctypes

from pcfa_amd.dropin._common import P, S, check, dense_gpu, lib


def forward(input1, input2, rbot1, rbot2, output, pad_size, kernel_size, max_displacement, stride1, stride2,
            corr_multiply):
    input1, input2 = input1.contiguous(), input2.contiguous()
    dense_gpu(input1, input2)
    B, C, H, W = input1.shape
    oc, oh, ow = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    check(lib().pcfa_flownet_corr_out_size(H, W, pad_size, kernel_size, max_displacement, stride1, stride2,
                                           ctypes.byref(oc), ctypes.byref(oh), ctypes.byref(ow)), "CUDA call failed: ")
    output.resize_(B, oc.value, oh.value, ow.value)
    check(lib().pcfa_flownet_corr_fwd(P(input1), P(input2), P(output), B, C, H, W, pad_size, kernel_size,
                                      max_displacement, stride1, stride2, S()), "CUDA call failed: ")
    return 1


def backward(input1, input2, rbot1, rbot2, grad_output, grad_input1, grad_input2, pad_size, kernel_size,
             max_displacement, stride1, stride2, corr_multiply):
    input1, input2, grad_output = input1.contiguous(), input2.contiguous(), grad_output.contiguous()
    dense_gpu(input1, input2, grad_output)
    B, C, H, W = input1.shape
    grad_input1.resize_(B, C, H, W)
    grad_input2.resize_(B, C, H, W)
    check(lib().pcfa_flownet_corr_bwd(P(input1), P(input2), P(grad_output), P(grad_input1), P(grad_input2), B, C, H,
                                      W, pad_size, kernel_size, max_displacement, stride1, stride2, S()),
          "CUDA call failed: ")
    return 1
