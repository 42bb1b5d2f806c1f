#!/usr/bin/env python3
"""Condense a rocprofv3 `*_kernel_stats.csv` into a short text table (kernel names truncated) that is
small enough to commit under profiles/.  Usage: summarize_rocprof.py <kernel_stats.csv> [top_n] > out.txt"""
import csv
import sys

OURS = ("corr_lookup", "scorr9", "sum_n", "conv3x3", "instnorm", "conv_fewin", "pwc_warp", "add_relu", "sepconv5", "lbfgs_", "gemm_f32_mfma", "f2ext", "splitk_reduce", "scorr_", "loss_", "box_", "deltas_", "gru_", "bias_relu", "relu_bwd", "null_kernel")


def main():
    path = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows = list(csv.DictReader(open(path)))
    raw = sum(float(r["TotalDurationNs"]) for r in rows)
    # MIOpen runs a slow generic kernel the FIRST time it meets a convolution configuration (immediate-mode
    # fallback while the tuned kernel is built); those one-off launches of the warm-up step are listed separately
    # and excluded from the percentages so that the table describes the steady state.
    naive = [r for r in rows if "naive_conv" in r["Name"]]
    rows = [r for r in rows if "naive_conv" not in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows:
        r["Percentage"] = "%.4f" % (100.0 * float(r["TotalDurationNs"]) / max(tot, 1))
    print("# source: %s" % path)
    print("# kernels: %d   device time: %.3f ms steady state + %.3f ms in %d MIOpen first-call fallback launches"
          % (len(rows), tot / 1e6, (raw - tot) / 1e6, sum(int(r["Calls"]) for r in naive)))
    fmt = "%-72s %8s %12s %12s %12s %12s %7s"
    print(fmt % ("name", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))

    def line(r):
        return fmt % (r["Name"][:72], r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6),
                      "%.2f" % (float(r["AverageNs"]) / 1e3), "%.2f" % (float(r["MinNs"]) / 1e3),
                      "%.2f" % (float(r["MaxNs"]) / 1e3), "%.2f" % float(r["Percentage"]))

    print("## top %d by total time" % top)
    for r in rows[:top]:
        print(line(r))
    print("## pcfa_amd HIP kernels (libpcfa_hip.so)")
    ours = [r for r in rows if any(k in r["Name"] for k in OURS)]
    for r in ours:
        print(line(r))
    print("# pcfa_amd kernels: %.3f ms = %.2f%% of device time" %
          (sum(float(r["TotalDurationNs"]) for r in ours) / 1e6,
           100 * sum(float(r["TotalDurationNs"]) for r in ours) / max(tot, 1)))


if __name__ == "__main__":
    main()
