#!/usr/bin/env python3
"""pcfa_gemm_f32 (the pyramid's fp32 MFMA GEMM core) at the shapes of the path, against torch.matmul (rocBLAS):
device time from the HIP activity tracer."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcfa_amd import hip_ops  # noqa: E402


def dev_time(fn, match=None):
    from torch.autograd import DeviceType
    from torch.profiler import ProfilerActivity, profile
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    return sum(ev.time_range.elapsed_us() for ev in prof.events() if ev.device_type == DeviceType.CUDA) / 10


def main():
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    Q = 7040
    cases = [("pyramid fwd  fmap1^T f2ext   M 7040 N 9616 K 256 ", (256, Q), (256, 9616), 1, 1, 1),
             ("sim  q k^T                  M 7040 N 7040 K 128 ", (Q, 128), (Q, 128), 0, 0, 1),
             ("attn v                      M 7040 N 128  K 7040", (Q, Q), (Q, 128), 0, 1, 8),
             ("attn^T g                    M 7040 N 128  K 7040", (Q, Q), (Q, 128), 1, 1, 8),
             ("d_attn [g|..] [v|..]^T      M 7040 N 7040 K 768 ", (Q, 768), (Q, 768), 0, 0, 1)]
    for name, sa, sb, akm, bkn, splits in cases:
        a = torch.randn(*sa, generator=g).to(dev)
        b = torch.randn(*sb, generator=g).to(dev)
        M = sa[1] if akm else sa[0]
        K = sa[0] if akm else sa[1]
        N = sb[1] if bkn else sb[0]
        t_mine = dev_time(lambda: hip_ops.gemm_f32(a, b, akm, bkn, splits=splits))
        at = a.t() if akm else a
        bt = b if bkn else b.t()
        t_lib = dev_time(lambda: torch.matmul(at, bt))
        fl = 2.0 * M * N * K
        print("%s  mine %7.1f us (%5.1f TFLOP/s)   rocBLAS %7.1f us (%5.1f TFLOP/s)" %
              (name, t_mine, fl / t_mine / 1e6, t_lib, fl / t_lib / 1e6))


if __name__ == "__main__":
    main()
