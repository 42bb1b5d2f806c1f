#!/usr/bin/env python3
"""Where does corr_lookup_fwd_kernel<4> spend its time?  Runs the in-kernel-stamped diagnostic build
(tools/dev/liblookup_dev.so) at the RAFT 55x128 shape and prints, per phase, the distribution over
workgroups of the cycles between stamps, the workgroup start skew and the span first-start -> last-end.
Diagnostic only: read SHARES, never quote this build's run time (cdna_hip_programming.md 7)."""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from pcfa_amd import hip_ops  # noqa: E402

DEV = "cuda"
NAMES = ["rt0", "start", "coords+bookkeeping", "window loads issued", "windows landed", "LDS written",
         "barrier", "blend (LDS reads)", "stores issued", "stores acked", "hwid", "rt1"]


def main():
    """usage: lookup_stamps.py stamps [variants]   -> phase tables of the stamped builds
              lookup_stamps.py time warm|cold [variants] -> N plain launches per variant (run under rocprofv3)"""
    what = sys.argv[1] if len(sys.argv) > 1 else "stamps"
    lib = ctypes.CDLL(os.path.join(HERE, "liblookup_dev.so"))
    P = ctypes.c_void_p
    lib.dev_lookup_fwd_var.argtypes = [ctypes.c_int] * 2 + [P, P, P] + [ctypes.c_int] * 4 + [P, P]
    slots = lib.dev_stamp_slots()
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
    out = torch.empty_like(want)
    nwg = (Q // 64) * 4
    stamps = torch.zeros(nwg * 9 * slots, dtype=torch.int64, device=DEV)
    junk = torch.empty(512 * 1024 * 1024 // 4, device=DEV)
    stream = torch.cuda.current_stream().cuda_stream

    def launch(var, stamped):
        rc = lib.dev_lookup_fwd_var(var, stamped, pyr.data_ptr(), coords.data_ptr(), out.data_ptr(), B, H, W, 4,
                                    stamps.data_ptr(), stream)
        assert rc == 0, rc

    if what == "time":
        mode = sys.argv[2]
        variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 2, 3]
        for var in variants:
            out.zero_()
            launch(var, 0)
            torch.cuda.synchronize()
            print("variant %d: max |diff| vs product %.3g, bit-equal %s" %
                  (var, float((out - want).abs().max()), torch.equal(out, want)))
        for rep in range(30):
            for var in variants:
                if mode == "cold":
                    junk.add_(1.0)
                launch(var, 0)
        torch.cuda.synchronize()
        return
    variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 2]
    for var, mode in [(v, m) for v in variants for m in ("warm", "cold")]:
        for rep in range(3):
            if mode == "cold":
                junk.add_(1.0)
            else:
                launch(var, 0)
            launch(var, 1)
            torch.cuda.synchronize()
        assert torch.equal(out, want), "stamped build changed the result"
        s = stamps.cpu().numpy().reshape(nwg, 9, slots).astype(np.int64)
        np.save(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "lookup_stamps_v%d_%s.npy" % (var, mode)), s)
        print("######## variant %d" % var)
        rt0, rt1 = s[:, :, 0], s[:, :, 11]
        t0 = rt0.min()
        print("== %s: %d workgroups x 9 waves; s_memrealtime tick = 10 ns" % (mode, nwg))
        print("span first wave start -> last wave end: %.2f us" % ((rt1.max() - t0) / 100.0))
        st = (rt0.min(axis=1) - t0) / 100.0
        en = (rt1.max(axis=1) - t0) / 100.0
        print("workgroup start (us after first): p10 %.2f p50 %.2f p90 %.2f max %.2f" %
              tuple(np.percentile(st, [10, 50, 90, 100])))
        print("workgroup end   (us after first): p10 %.2f p50 %.2f p90 %.2f max %.2f" %
              tuple(np.percentile(en, [10, 50, 90, 100])))
        print("workgroup lifetime us: p10 %.2f p50 %.2f p90 %.2f max %.2f" %
              tuple(np.percentile(en - st, [10, 50, 90, 100])))
        print("wave start skew inside a workgroup (us): p50 %.2f p90 %.2f max %.2f" %
              tuple(np.percentile((rt0.max(axis=1) - rt0.min(axis=1)) / 100.0, [50, 90, 100])))
        print("%-26s %8s %8s %8s %8s   (shader cycles between consecutive stamps, all waves)" %
              ("phase", "p10", "p50", "p90", "max"))
        for k in range(2, 10):
            d = (s[:, :, k] - s[:, :, k - 1]).ravel()
            print("%-26s %8d %8d %8d %8d" % ((NAMES[k],) + tuple(np.percentile(d, [10, 50, 90, 100]).astype(int))))
        tot = (s[:, :, 9] - s[:, :, 1]).ravel()
        print("%-26s %8d %8d %8d %8d" % (("stamp 1 -> 9",) + tuple(np.percentile(tot, [10, 50, 90, 100]).astype(int))))
        life = (rt1 - rt0).ravel() / 100.0
        print("clock estimate: %.2f GHz" % (np.median(tot / np.maximum(life, 1e-9)) / 1e3))
        # per level
        wl = np.arange(nwg) // (Q // 64)
        for lv in range(4):
            m = wl == lv
            print("level %d: lifetime p50 %.2f us, windows landed p50 %d cyc, start p50 %.2f us" %
                  (lv, np.median((en - st)[m]), np.median((s[m, :, 4] - s[m, :, 3])), np.median(st[m])))
        # placement
        xcc = (s[:, 0, 10] >> 32) & 0xF
        hw = s[:, 0, 10] & 0xFFFFFFFF
        key = (xcc << 8) | ((hw >> 8) & 0xFF)   # HW_ID[15:8] = CU_ID, SH_ID, SE_ID
        u, c = np.unique(key, return_counts=True)
        print("placement: %d distinct (xcc,se,sh,cu) hold the %d workgroups; per-CU count histogram %s" %
              (len(u), nwg, dict(zip(*np.unique(c, return_counts=True)))))


if __name__ == "__main__":
    main()
