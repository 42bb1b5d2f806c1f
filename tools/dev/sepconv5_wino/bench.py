#!/usr/bin/env python3
"""PROTOTYPE: 1x5 convolution as 1-D Winograd F(2,5) against the product's sepconv5 kernel and torch, RAFT gate shape.
Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC proto.hip -o libwino15.so (build.sh)."""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(HERE))))
from pcfa_amd import hip_ops  # noqa: E402
from tools.dev.bench_conv3x3 import device_us  # noqa: E402

lib = ctypes.CDLL(os.path.join(HERE, os.environ.get("WINO_LIB", "libwino15.so")))
lib.wino15_packed_floats.restype = ctypes.c_longlong
P = ctypes.c_void_p
dev = "cuda"
C, N, H, W = 256, 256, int(os.environ.get("WINO_H", "55")), 128
g = torch.Generator().manual_seed(0)
x = torch.randn(1, C, H, W, generator=g).to(dev)
w = (torch.randn(N, C, 1, 5, generator=g) / (5 * C) ** .5).to(dev)
packed = torch.empty(lib.wino15_packed_floats(C, N), device=dev)
s = torch.cuda.current_stream().cuda_stream
assert lib.wino15_pack(P(w.data_ptr()), P(packed.data_ptr()), C, N, P(s)) == 0
out = torch.empty(1, N, H, W, device=dev)
run = lambda: lib.wino15_run(P(x.data_ptr()), P(packed.data_ptr()), P(out.data_ptr()), C, N, H, W, P(s))  # noqa: E731
assert run() == 0
torch.cuda.synchronize()
ref = F.conv2d(x.double(), w.double(), None, padding=(0, 2))
got32 = F.conv2d(x, w, None, padding=(0, 2))
print("F(2,5) vs fp64: max %.2e rms %.2e   torch fp32 vs fp64: max %.2e rms %.2e" % (
    (out.double() - ref).abs().max(), (out.double() - ref).pow(2).mean().sqrt(), (got32.double() - ref).abs().max(),
    (got32.double() - ref).pow(2).mean().sqrt()))
t, parts = device_us(run)
print("F(2,5) prototype: %.1f us (%.1f GFLOP direct-equivalent -> %.1f TFLOP/s)" % (t, 2 * 5 * C * N * H * W * 1e-9, 2 * 5 * C * N * H * W / t * 1e-6))
h, m = x[:, :128].contiguous(), x[:, 128:].contiguous()
t2, parts2 = device_us(lambda: hip_ops.sepconv5(h, m, w))
print("sepconv5 (product, direct implicit GEMM): %.1f us" % t2)
for k, v in parts2.items():
    print("    %.1f us %s" % (v, k[:70]))

# ---- 5x1 (vertical) form ----
wv_ = (torch.randn(N, C, 5, 1, generator=g) / (5 * C) ** .5).to(dev)
packed_v = torch.empty(lib.wino15_packed_floats(C, N), device=dev)
assert lib.wino15_pack(P(wv_.data_ptr()), P(packed_v.data_ptr()), C, N, P(s)) == 0     # same packing: taps along the axis
out_v = torch.empty(1, N, H, W, device=dev)
run_v = lambda: lib.wino51_run(P(x.data_ptr()), P(packed_v.data_ptr()), P(out_v.data_ptr()), C, N, H, W, P(s))  # noqa: E731
assert run_v() == 0
torch.cuda.synchronize()
ref_v = F.conv2d(x.double(), wv_.double(), None, padding=(2, 0))
print("5x1 F(2,5) vs fp64: max %.2e rms %.2e" % ((out_v.double() - ref_v).abs().max(), (out_v.double() - ref_v).pow(2).mean().sqrt()))
t, _ = device_us(run_v)
print("5x1 F(2,5) prototype (K split): %.1f us" % t)
t2, _ = device_us(lambda: hip_ops.sepconv5(h, m, wv_))
print("5x1 sepconv5 (product): %.1f us" % t2)
