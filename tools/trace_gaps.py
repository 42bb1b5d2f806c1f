#!/usr/bin/env python3
"""Steady-state analysis of a rocprofv3 kernel trace: GPU busy fraction, inter-kernel gaps and the time split by
kernel family over the LAST `window_s` seconds of the run (after warm-up).  Usage: trace_gaps.py <kernel_trace.csv> [window_s]"""
import csv
import sys


def family(n):
    if 'naive_conv' in n: return 'miopen naive (first-call fallback)'
    if 'Cijk' in n: return 'rocblas gemm (1x1 / im2col convs)'
    if 'miopenSp3AsmConv' in n or 'igemm' in n or 'ck::' in n or n.startswith('_ZN2ck'): return 'miopen conv'
    if 'Im2d2Col' in n or 'Col2Im' in n: return 'im2col/col2im'
    if 'transpose' in n or 'SubTensorOp' in n: return 'miopen aux'
    if 'anonymous namespace' in n and any(k in n for k in ('corr_', 'gemm_f32', 'f2ext', 'splitk', 'loss_', 'box_', 'deltas_', 'gru_', 'null_kernel', 'scorr_')): return 'pcfa_amd'
    if 'elementwise' in n or 'Cat' in n or 'copy' in n.lower() or 'fill' in n.lower(): return 'torch elementwise/copy/cat'
    if 'batch_norm' in n or 'BatchNorm' in n: return 'norms'
    if 'reduce' in n or 'rocblas_dot' in n or 'rocblas_reduction' in n: return 'reductions'
    return 'other'


def main():
    path = sys.argv[1]
    window = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    rows = []
    for r in csv.DictReader(open(path)):
        if "naive_conv" in r["Kernel_Name"]:
            continue  # MIOpen's one-off first-call fallback kernels of the warm-up step
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # the busiest `window` seconds of the run = the timed steps (the tail of a bench.py run is the CPU baseline)
    w_ns = int(window * 1e9)
    starts = [r[0] for r in rows]
    import bisect
    best, best_t0 = -1, rows[0][0]
    t = rows[0][0]
    while t + w_ns <= rows[-1][1]:
        i, j = bisect.bisect_left(starts, t), bisect.bisect_left(starts, t + w_ns)
        busy_t = sum(e - s for s, e, _ in rows[i:j])
        if busy_t > best:
            best, best_t0 = busy_t, t
        t += w_ns // 4
    sel = [r for r in rows if best_t0 <= r[0] < best_t0 + w_ns]
    busy = sum(e - s for s, e, _ in sel)
    span = sel[-1][1] - sel[0][0]
    gaps = [sel[i + 1][0] - sel[i][1] for i in range(len(sel) - 1)]
    pos = [g for g in gaps if g > 0]
    print("window %.3f s: %d kernels, busy %.1f%%, mean kernel %.2f us, mean gap %.2f us, gaps>20us: %d (%.1f ms)" %
          (span / 1e9, len(sel), 100.0 * busy / span, busy / len(sel) / 1e3, sum(pos) / max(len(pos), 1) / 1e3,
           sum(1 for g in pos if g > 20000), sum(g for g in pos if g > 20000) / 1e6))
    fam = {}
    for s, e, n in sel:
        a = fam.setdefault(family(n), [0, 0])
        a[0] += e - s
        a[1] += 1
    for k, (t, c) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print("  %-38s %8.2f ms (%5.1f%% of span) %7d launches  avg %6.1f us" % (k, t / 1e6, 100.0 * t / span, c, t / c / 1e3))


if __name__ == "__main__":
    main()
