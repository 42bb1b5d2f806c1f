"""`resample2d_cuda` on libpcfa_hip.so (models/FlowNet/resample2d_package/resample2d_cuda.cc:6-24; kernels
resample2d_kernel.cu:16-72,75-201): caller-allocated tensors, returns 1."""
from pcfa_amd.dropin._common import P, S, check, dense_gpu, lib


def forward(input1, input2, output, kernel_size, bilinear=True):
    dense_gpu(input1, input2, output)
    B, C, iH, iW = input1.shape
    H, W = input2.shape[-2:]
    check(lib().pcfa_resample2d_fwd(P(input1), P(input2), P(output), B, C, iH, iW, H, W, kernel_size, int(bilinear),
                                    S()))
    return 1


def backward(input1, input2, grad_output, grad_input1, grad_input2, kernel_size, bilinear=True):
    grad_output = grad_output.contiguous()
    dense_gpu(input1, input2, grad_output, grad_input1, grad_input2)
    B, C, iH, iW = input1.shape
    H, W = input2.shape[-2:]
    check(lib().pcfa_resample2d_bwd(P(input1), P(input2), P(grad_output), P(grad_input1), P(grad_input2), B, C, iH, iW,
                                    H, W, kernel_size, int(bilinear), S()))   # clears grad_input1 itself
    return 1
