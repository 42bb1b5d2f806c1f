#!/usr/bin/env python3
"""Micro-benchmark of every pcfa_amd HIP kernel at BASELINE shapes: mean launch time from HIP events
around back-to-back launches (L2/MALL-warm), algorithmic GB/s (SURVEY.md 8d byte counts) or TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pcfa_amd import hip_ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=50, warm=5, flush=None):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if flush is None:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3 / iters
    tot = 0.0
    for _ in range(iters):
        flush()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        tot += s.elapsed_time(e) * 1e3
    return tot / iters


def main():
    B, D, H, W = 1, 256, 55, 128
    Q = H * W
    g = torch.Generator().manual_seed(0)
    f1 = torch.randn(B, D, H, W, generator=g).to(DEV).requires_grad_(True)
    f2 = torch.randn(B, D, H, W, generator=g).to(DEV).requires_grad_(True)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    coords = (torch.stack([xs, ys], 0).float()[None] + 3 * torch.randn(B, 2, H, W, generator=g)).to(DEV)
    junk = torch.empty(512 * 1024 * 1024 // 4, device=DEV)  # > MALL: evicts L2 + Infinity Cache

    def flush():
        junk.add_(1.0)

    t = timeit(lambda: hip_ops.CorrBlock(f1.detach(), f2.detach()), iters=20)
    print("corr build (f2ext + pyramid GEMM)  %9.1f us   %.1f TFLOP/s" % (t, 2 * Q * 9600 * D / t / 1e6))
    blk = hip_ops.CorrBlock(f1, f2)
    nb = Q * 4 * 100 * 4 + Q * 8 + Q * 324 * 4
    t = timeit(lambda: blk(coords))
    print("corr_lookup_fwd warm               %9.1f us   %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" %
          (t, nb / t / 1e3, nb / t / 1e3 / 80))
    t = timeit(lambda: blk(coords), iters=20, flush=flush)
    print("corr_lookup_fwd cold (L2+MALL flushed) %5.1f us   %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" %
          (t, nb / t / 1e3, nb / t / 1e3 / 80))
    out = blk(coords)
    go = torch.randn_like(out)
    st = blk._state
    st.dpyr = torch.zeros_like(st.pyr)
    from pcfa_amd.hip_ops import _call, _ptr
    nbb = Q * 324 * 4 + Q * 8 + 2 * Q * 4 * 100 * 4
    t = timeit(lambda: _call("pcfa_corr_lookup_bwd", _ptr(st.dpyr), _ptr(coords), _ptr(go), B, H, W, 4, 4))
    print("corr_lookup_bwd warm               %9.1f us   %.0f GB/s algorithmic" % (t, nbb / t / 1e3))

    def full_bwd():
        o = blk(coords)
        o.backward(go, retain_graph=True)
    t = timeit(full_bwd, iters=10)
    print("lookup fwd + lookup bwd + pyramid bwd (2 GEMMs) %9.1f us" % t)

    # SepConvGRU gate convolutions at 55x128 (RAFT: [h | motion] = 128+128 channels), against the library conv
    import torch.nn.functional as F
    for name, cout, cb in (("z|r", 256, 128), ("q", 128, 128), ("GMA z|r", 256, 256)):
        for kh, kw in ((1, 5), (5, 1)):
            a = torch.randn(1, 128, H, W, device=DEV)
            b = torch.randn(1, cb, H, W, device=DEV)
            wgt = torch.randn(cout, 128 + cb, kh, kw, device=DEV) * 0.03
            gf = 2 * 5 * (128 + cb) * cout * Q
            t = timeit(lambda: hip_ops.sepconv5(a, b, wgt))
            tl = timeit(lambda: F.conv2d(torch.cat([a, b], 1), wgt, None, padding=(kh // 2, kw // 2)))
            print("sepconv5 %-8s %dx%d  %7.1f us  %5.1f TFLOP/s   | cat + library conv2d %7.1f us" %
                  (name, kh, kw, t, gf / t / 1e6, tl))

    # PWC cost volume at KITTI level shapes
    for (C, h, w) in ((196, 6, 20), (128, 12, 40), (96, 24, 80), (64, 48, 160), (32, 96, 320)):
        a = torch.randn(1, C, h, w, device=DEV, requires_grad=True)
        b = torch.randn(1, C, h, w, device=DEV, requires_grad=True)
        t = timeit(lambda: hip_ops.spatial_correlation_sample(a.detach(), b.detach(), 1, 9, 1))
        byt = 2 * C * h * w * 4 + 81 * h * w * 4
        o = hip_ops.spatial_correlation_sample(a, b, 1, 9, 1)
        gg = torch.randn_like(o)
        tb = timeit(lambda: o.backward(gg, retain_graph=True), iters=20)
        print("spatial_corr %3dx%3dx%3d  fwd %7.1f us (%.0f GB/s)   bwd %7.1f us (%.0f GB/s)" %
              (C, h, w, t, byt / t / 1e3, tb, (byt + 2 * C * h * w * 4) / tb / 1e3))

    # attack math at 440x1024
    n = 3 * 440 * 1024
    img = torch.rand(1, 3, 440, 1024, device=DEV)
    w_ = torch.randn(1, 3, 440, 1024, device=DEV, requires_grad=True)
    t = timeit(lambda: hip_ops.box_transform(w_.detach(), None, True, 1e-7, 255.))
    print("box_transform fwd                  %9.1f us   %.0f GB/s" % (t, 2 * n * 4 / t / 1e3))
    t = timeit(lambda: hip_ops._ExtractDeltas.apply(w_.detach(), img, True, 1e-7))
    print("extract_deltas fwd                 %9.1f us   %.0f GB/s" % (t, 3 * n * 4 / t / 1e3))
    flow = torch.randn(1, 2, 440, 1024, device=DEV)[..., 2:438, :]
    tgt = torch.zeros(1, 2, 436, 1024, device=DEV)
    d1 = 0.01 * torch.randn(1, 3, 440, 1024, device=DEV)
    d2 = 0.01 * torch.randn(1, 3, 440, 1024, device=DEV)
    t = timeit(lambda: hip_ops.loss_delta_constraint(flow, tgt, d1, d2, None, 0.005, 5e5, "aee"))
    print("flow_loss fwd (2 kernels)          %9.1f us   %.0f GB/s" % (t, (2 * 2 * 436 * 1024 * 4 + 2 * n * 4) / t / 1e3))


if __name__ == "__main__":
    main()
