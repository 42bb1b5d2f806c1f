#!/usr/bin/env python3
"""Where does a conv3x3 (Winograd) loop iteration spend its cycles?  Needs the -DPCFA_C3_STAMPS build
(PCFA_HIP_LIB=pcfa_amd/lib/libpcfa_hip_stamps.so).  Shares only -- the stamped build is slower."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcfa_amd import _hip, hip_ops  # noqa: E402

lib = _hip.load()
lib.dev_c3_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 8)()
names = ["", "MFMA phase (16 MFMAs + next patch -> LDS + raw loads of chunk c+2 in their shadow)", "(empty)",
         "issue U loads of chunk c+2", "barrier"]
for B, K, N, H, W in [(1, 256, 192, 55, 128), (1, 256, 126, 55, 128), (1, 128, 256, 55, 128), (2, 64, 64, 220, 512)]:
    x = torch.randn(B, K, H, W, device="cuda")
    w = torch.randn(N, K, 3, 3, device="cuda") / (9 * K) ** .5
    b = torch.randn(N, device="cuda")
    for _ in range(3):
        hip_ops.conv3x3(x, w, b, True)
    torch.cuda.synchronize()
    lib.dev_c3_stamps(buf, 1)
    for _ in range(5):
        hip_ops.conv3x3(x, w, b, True)
    torch.cuda.synchronize()
    lib.dev_c3_stamps(buf, 1)
    n = max(buf[0], 1)
    tot = sum(buf[k] for k in range(1, 5))
    print("conv3x3 B%d %d->%d %dx%d: %.0f cycles per chunk (wave 0 of every workgroup)" % (B, K, N, H, W, tot / n))
    for k in range(1, 5):
        print("   %-62s %7.0f cycles  %5.1f %%" % (names[k], buf[k] / n, 100.0 * buf[k] / tot))
