#!/usr/bin/env python3
"""Phase timestamps of corr_lookup_convc1_fwd_kernel / _bwd_kernel (needs a -DPCFA_LC_DBG_BUILD=8 build selected with
PCFA_HIP_LIB): per workgroup, shader-clock ticks from kernel entry to each phase boundary.  `--bwd` for the backward."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcfa_amd import hip_ops  # noqa: E402

dev = torch.device("cuda")
H, W, D = 55, 128, 256
g = torch.Generator().manual_seed(0)
f1 = torch.randn(1, D, H, W, generator=g).to(dev)
f2 = torch.randn(1, D, H, W, generator=g).to(dev)
w = (torch.randn(256, 324, 1, 1, generator=g) / 18).to(dev)
b = torch.zeros(256, device=dev)
ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
coords = (torch.stack([xs, ys], 0).float()[None] + 3 * torch.randn(1, 2, H, W, generator=g)).to(dev)
if "--bwd" in sys.argv:
    blk = hip_ops.CorrBlock(f1, f2)
    st = blk._state
    go = torch.randn(1, 256, H, W, generator=g).to(dev)
    with torch.no_grad():
        out = blk.lookup_conv_relu(coords, w, b, True)
    packed = hip_ops._convc1_packed(w)
    dpyr = torch.zeros_like(st.pyr)
    for _ in range(3):
        dpyr.zero_()
        hip_ops._call("pcfa_lookup_convc1_bwd", hip_ops._ptr(dpyr), hip_ops._ptr(coords), hip_ops._ptr(packed),
                      hip_ops._ptr(out), hip_ops._ptr(go), st.B, st.H, st.W, st.L, st.r, 256, 1)
    torch.cuda.synchronize()
    t = dpyr.view(-1, st.slab)[::32, :8].cpu()
    names = ["entry", "geometry done", "gradient tile staged", "GEMM done", "level 0 scattered", "level 1 scattered",
             "level 2 scattered", "level 3 scattered"]
    for k in range(8):
        c = t[:, k]
        print("%-22s median %8.0f  min %8.0f  max %8.0f cycles" % (names[k], c.median(), c.min(), c.max()))
    sys.exit(0)
blk = hip_ops.CorrBlock(f1, f2)
with torch.no_grad():
    for _ in range(3):
        out = blk.lookup_conv_relu(coords, w, b, False)
torch.cuda.synchronize()
n = (H * W + 31) // 32
t = out.view(256, H * W)[:8, ::32].t().cpu()
names = ["entry", "windows L3 staged", "prologue done", "level 3 done", "level 2 done", "level 1 done", "level 0 done",
         "stores issued"]
for k in range(8):
    c = t[:, k]
    print("%-20s median %8.0f  min %8.0f  max %8.0f cycles" % (names[k], c.median(), c.min(), c.max()))
