#!/usr/bin/env python3
"""Where do the correlation-lookup kernels spend their time?  (RAFT shape 55x128, radius 4, 4 levels.)

  lookup_stamps.py stamps          phase tables of the in-kernel-stamped diagnostic builds (tools/dev/liblookup_dev.so):
                                   per phase the distribution over waves of the cycles between stamps, workgroup start
                                   skew, residency per CU.  Read SHARES, never quote this build's run time.
  lookup_stamps.py time warm|cold  30 plain launches of the product forward and backward kernels; run it under
                                   `rocprofv3 --kernel-trace --stats` to read their durations (cold = L2 and the
                                   Infinity Cache flushed by a 512 MiB sweep before every launch)."""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from pcfa_amd import hip_ops  # noqa: E402
from pcfa_amd.hip_ops import _call, _ptr  # noqa: E402

DEV = "cuda"
FWD = ["rt0", "start", "coords landed", "piece loads issued", "pieces landed", "LDS image written", "barrier",
       "blend (LDS reads)", "stores issued", "stores acked", "hwid", "rt1"]
BWD = ["rt0", "start", "coords landed", "dpyr loads issued", "image rows written", "barrier", "dpyr pieces landed",
       "add + stores issued", "stores acked", "-", "hwid", "rt1"]


def report(s, names, nwg, Q, label):
    rt0, rt1 = s[:, :, 0], s[:, :, 11]
    t0 = rt0.min()
    print("== %s: %d workgroups x %d waves; s_memrealtime tick = 10 ns" % (label, nwg, s.shape[1]))
    print("span first wave start -> last wave end: %.2f us" % ((rt1.max() - t0) / 100.0))
    st = (rt0.min(axis=1) - t0) / 100.0
    en = (rt1.max(axis=1) - t0) / 100.0
    print("workgroup start (us after first): p10 %.2f p50 %.2f p90 %.2f max %.2f" %
          tuple(np.percentile(st, [10, 50, 90, 100])))
    print("workgroup lifetime us: p10 %.2f p50 %.2f p90 %.2f max %.2f" %
          tuple(np.percentile(en - st, [10, 50, 90, 100])))
    print("%-26s %8s %8s %8s %8s   (shader cycles between consecutive stamps, all waves)" %
          ("phase", "p10", "p50", "p90", "max"))
    for k in range(2, 10):
        if names[k] == "-":
            continue
        d = (s[:, :, k] - s[:, :, k - 1]).ravel()
        print("%-26s %8d %8d %8d %8d" % ((names[k],) + tuple(np.percentile(d, [10, 50, 90, 100]).astype(int))))
    tot = (s[:, :, 9] - s[:, :, 1]).ravel()
    print("%-26s %8d %8d %8d %8d" % (("stamp 1 -> 9",) + tuple(np.percentile(tot, [10, 50, 90, 100]).astype(int))))
    xcc = (s[:, 0, 10] >> 32) & 0xF
    hw = s[:, 0, 10] & 0xFFFFFFFF
    key = (xcc << 8) | ((hw >> 8) & 0xFF)  # HW_ID[15:8] = CU_ID, SH_ID, SE_ID
    best = 0
    for k in np.unique(key):
        ev = sorted([(st[g], 1) for g in np.nonzero(key == k)[0]] + [(en[g], -1) for g in np.nonzero(key == k)[0]])
        cur = 0
        for _, dlt in ev:
            cur += dlt
            best = max(best, cur)
    print("placement: %d CUs hold the %d workgroups, at most %d resident on one CU at a time" %
          (len(np.unique(key)), nwg, best))


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "stamps"
    B, D, H, W = 1, 256, 55, 128
    Q = H * W
    g = torch.Generator().manual_seed(0)
    f1 = torch.randn(B, D, H, W, generator=g).to(DEV)
    f2 = torch.randn(B, D, H, W, generator=g).to(DEV)
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    coords = (torch.stack([xs, ys], 0).float()[None] + 3 * torch.randn(B, 2, H, W, generator=g)).to(DEV)
    blk = hip_ops.CorrBlock(f1, f2)
    want = blk(coords)
    pyr = blk._state.pyr
    go = torch.randn(want.shape, generator=g).to(DEV)
    dpyr = torch.zeros_like(pyr)
    junk = torch.empty(512 * 1024 * 1024 // 4, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream

    def product_fwd(out):
        _call("pcfa_corr_lookup_fwd", _ptr(pyr), _ptr(coords), _ptr(out), B, H, W, 4, 4)

    def product_bwd(dst):
        _call("pcfa_corr_lookup_bwd", _ptr(dst), _ptr(coords), _ptr(go), B, H, W, 4, 4)

    if what == "stores":  # A/B of the forward kernel's store flavour (run under rocprofv3)
        mode = sys.argv[2]
        lib = ctypes.CDLL(os.environ.get("PCFA_LOOKUP_DEV_LIB") or os.path.join(HERE, "liblookup_dev.so"))
        P = ctypes.c_void_p
        lib.dev_lookup_fwd_store.argtypes = [ctypes.c_int, P, P, P] + [ctypes.c_int] * 4 + [P]
        out = torch.empty_like(want)
        for rep in range(30):
            for m in (0, 1, 2):
                if mode == "cold":
                    junk.add_(1.0)
                assert lib.dev_lookup_fwd_store(m, pyr.data_ptr(), coords.data_ptr(), out.data_ptr(), B, H, W, 4,
                                                stream) == 0
        torch.cuda.synchronize()
        assert float((out - want).abs().max()) < 1e-5
        return
    if what == "live":  # dispatch-attached events (what bench.py reports) on the same back-to-back launches
        mode = sys.argv[2]
        out = torch.empty_like(want)
        for flags in (0x20000000, 0):
            timer = hip_ops.DispatchTimer()
            timer.EVENT_FLAGS = flags
            for rep in range(30):
                for fn, buf in ((product_fwd, out), (product_bwd, dpyr)):
                    if mode == "cold":
                        junk.add_(1.0)
                    if rep >= 5:
                        hip_ops.set_dispatch_timer(timer)
                    fn(buf)
                    hip_ops.set_dispatch_timer(None)
            print(mode, "event flags 0x%x:" % flags, {k: (round(v[0], 2), v[1]) for k, v in timer.summary().items()})
        return
    if what == "time":
        mode = sys.argv[2]
        out = torch.empty_like(want)
        for rep in range(30):
            for fn, buf in ((product_fwd, out), (product_bwd, dpyr)):
                if mode == "cold":
                    junk.add_(1.0)
                fn(buf)
        torch.cuda.synchronize()
        return

    lib = ctypes.CDLL(os.environ.get("PCFA_LOOKUP_DEV_LIB") or os.path.join(HERE, "liblookup_dev.so"))
    P = ctypes.c_void_p
    lib.dev_lookup_fwd_stamped.argtypes = [P, P, P] + [ctypes.c_int] * 4 + [P, P]
    lib.dev_lookup_bwd_stamped.argtypes = [P, P, P] + [ctypes.c_int] * 4 + [P, P]
    slots, nw, qb = lib.dev_stamp_slots(), lib.dev_waves(), lib.dev_qb()
    nwg = (Q // qb) * 4
    stamps = torch.zeros(nwg * nw * slots, dtype=torch.int64, device=DEV)
    out = torch.empty_like(want)
    dref = torch.zeros_like(pyr)
    product_bwd(dref)
    for direction in ("fwd", "bwd"):
        for mode in ("warm", "cold"):
            for rep in range(3):
                if direction == "bwd":
                    dpyr.zero_()
                if mode == "cold":
                    junk.add_(1.0)
                elif direction == "fwd":
                    product_fwd(out)
                else:
                    product_bwd(dpyr)
                    dpyr.zero_()
                if direction == "fwd":
                    rc = lib.dev_lookup_fwd_stamped(pyr.data_ptr(), coords.data_ptr(), out.data_ptr(), B, H, W, 4,
                                                    stamps.data_ptr(), stream)
                else:
                    rc = lib.dev_lookup_bwd_stamped(dpyr.data_ptr(), coords.data_ptr(), go.data_ptr(), B, H, W, 4,
                                                    stamps.data_ptr(), stream)
                assert rc == 0, rc
                torch.cuda.synchronize()
            got, ref = (out, want) if direction == "fwd" else (dpyr, dref)
            err = float((got - ref).abs().max())
            print("stamped vs product: max |diff| %.3g (max |ref| %.3g)" % (err, float(ref.abs().max())))
            assert err <= 1e-5 * float(ref.abs().max()), "stamped build changed the result"
            s = stamps.cpu().numpy().reshape(nwg, nw, slots).astype(np.int64)
            report(s, FWD if direction == "fwd" else BWD, nwg, Q, "%s %s" % (direction, mode))


if __name__ == "__main__":
    main()
