"""`channelnorm_cuda` on libpcfa_hip.so (models/FlowNet/channelnorm_package/channelnorm_cuda.cc:6-25; kernels
channelnorm_kernel.cu:18-60,63-96): caller-allocated tensors, returns 1."""
from pcfa_amd.dropin._common import P, S, check, dense_gpu, lib


def forward(input1, output, norm_deg):
    dense_gpu(input1, output)
    B, C, H, W = input1.shape
    check(lib().pcfa_channelnorm_fwd(P(input1), P(output), B, C, H * W, norm_deg, S()))
    return 1


def backward(input1, output, grad_output, grad_input1, norm_deg):
    grad_output = grad_output.contiguous()
    dense_gpu(input1, output, grad_output, grad_input1)
    B, C, H, W = input1.shape
    check(lib().pcfa_channelnorm_bwd(P(input1), P(output), P(grad_output), P(grad_input1), B, C, H * W, norm_deg, S()))
    return 1
