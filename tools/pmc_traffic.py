#!/usr/bin/env python3
"""Per-launch HBM traffic of the pcfa_amd kernels from two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE).

Corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB; on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide coalesced stream (16 B/lane loads -- what these kernels issue), so it is doubled;
WRITE_SIZE is taken as is.  Prints JSON: kernel -> {launches, fetch_bytes, write_bytes, traffic_bytes} (means)."""
import csv
import glob
import json
import os
import sys

OURS = ("corr_lookup_convc1_fwd", "corr_lookup_convc1_bwd", "corr_lookup_fwd", "corr_lookup_bwd", "scorr9_fwd",
        "scorr9_bwd", "sum_n_kernel", "gemm_f32_mfma", "f2ext_", "scorr_", "loss_partial", "box_fwd",
        "deltas_fwd", "instnorm_stats_kernel<false>", "instnorm_stats_kernel<true>", "instnorm_apply_kernel<false>",
        "instnorm_apply_kernel<true>", "add_relu_kernel", "gru_gates_fwd", "gru_update_fwd", "pwc_warp_fwd",
        "pwc_warp_bwd_det_lds", "pwc_warp_bwd_det", "pwc_warp_finish", "zero_ll_max", "conv3x3_winograd_kernel",
        "sc5_wino_kernel", "corr_pyramid_pool_gemm", "instnorm_plane_kernel", "gram_pass_kernel", "gram_direction_kernel",
        "conv_s2_fwd_kernel", "conv_s2_bwd_kernel")
PER_GRID = ("scorr9_fwd", "scorr9_bwd", "pwc_warp_fwd", "pwc_warp_bwd_det_lds", "conv3x3_winograd_kernel", "sc5_wino_kernel",
            "instnorm_plane_kernel", "conv_s2_fwd_kernel", "conv_s2_bwd_kernel")   # one row per launch shape (= PWC-Net level)


def collect(folder, counter):
    out = {}
    for path in glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            key = next((k for k in OURS if k in name), None)
            if key is None:
                continue
            if key == "gemm_f32_mfma":
                key = name[name.index("gemm_f32_mfma"):].split("(")[0]
            if key in PER_GRID and row.get("Grid_Size"):   # template arguments + launch shape name the PWC-Net level
                targs = name[name.index(key) + len(key):].split("(")[0].replace("_kernel", "")
                key = "%s%s grid %s wg %s" % (key, targs, row["Grid_Size"], row.get("Workgroup_Size", "?"))
            out.setdefault(key, []).append(float(row["Counter_Value"]))
    return out


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, [])
        w = write.get(k, [])
        fb = 2.0 * 1024.0 * sum(f) / len(f) if f else None  # x2: gfx950 FETCH_SIZE correction
        wb = 1024.0 * sum(w) / len(w) if w else None
        res[k] = {"launches": max(len(f), len(w)), "fetch_bytes": fb, "write_bytes": wb,
                  "traffic_bytes": (fb or 0) + (wb or 0),
                  "note": "FETCH_SIZE KiB x2 (gfx950 wide-read correction) + WRITE_SIZE KiB, mean per launch"}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
