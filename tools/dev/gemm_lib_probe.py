#!/usr/bin/env python3
"""Library fp32 GEMM vs pcfa_gemm_f32 on the three correlation-pyramid products at 55x128, D = 256 (device time per
launch from the HIP activity tracer)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcfa_amd import hip_ops  # noqa: E402

dev = torch.device("cuda")
Q, D, S = 7040, 256, 9616
g = torch.Generator().manual_seed(0)
f1 = torch.randn(D, Q, generator=g).to(dev)        # fmap1 [D][Q]
f2e = torch.randn(D, S, generator=g).to(dev)       # f2ext [D][S]
dp = torch.randn(Q, S, generator=g).to(dev)        # dpyr [Q][S]
out = torch.empty(Q, S, device=dev)
cases = {
    "fwd  pyr[Q,S] = f1^T f2ext      (lib addmm)": lambda: torch.addmm(out, f1.t(), f2e, beta=0, alpha=1 / 16, out=out),
    "bwd a dfmap1[D,Q] = f2ext dpyr^T (lib)": lambda: torch.matmul(f2e, dp.t()),
    "bwd b df2ext[D,S] = f1 dpyr      (lib)": lambda: torch.matmul(f1, dp),
    "fwd  (pcfa_gemm_f32)": lambda: hip_ops.gemm_f32(f1, f2e, 1, 1, alpha=1 / 16),
    "bwd a (pcfa_gemm_f32, 8 splits)": lambda: hip_ops.gemm_f32(f2e, dp, 0, 0, splits=8),
    "bwd b (pcfa_gemm_f32, 8 splits)": lambda: hip_ops.gemm_f32(f1, dp, 0, 1, splits=8),
}
from torch.autograd import DeviceType
from torch.profiler import ProfilerActivity, profile
for name, fn in cases.items():
    try:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
        acc = {}
        for ev in prof.events():
            if ev.device_type == DeviceType.CUDA:
                acc.setdefault(ev.name[:70], []).append(ev.time_range.elapsed_us())
        tot = sum(sum(v) for v in acc.values()) / 10
        print("%-46s %8.1f us  %s" % (name, tot, "; ".join("%s x%d %.1f" % (k[:40], len(v) // 10, sum(v) / len(v)) for k, v in acc.items())))
    except Exception as e:  # noqa: BLE001
        print(name, "FAILED", repr(e)[:200])
