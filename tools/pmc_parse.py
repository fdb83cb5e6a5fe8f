#!/usr/bin/env python3
"""Turns the two rocprofv3 counter passes of tools/pmc_run.py into HBM bytes per launch.

Method (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are
collected in separate passes (they do not fit one pass), both are in KiB; on gfx950
FETCH_SIZE under-reports wide coalesced reads, so each counter is CALIBRATED on the
stream-copy kernels of the same run, whose byte counts are known, with the copy whose
per-lane access width matches the kernel's dominant width (8 B/lane for CAAR's scalar
fields) — the correction factor is printed and stored beside the result.
"""
import csv
import glob
import json
import os
import sys


def load(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                rows.append((r["Kernel_Name"], float(r["Counter_Value"])))
    return rows


def mean(xs):
    xs = list(xs)
    return sum(xs) / len(xs) if xs else float("nan")


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    copy_bytes = float(sys.argv[3]) if len(sys.argv) > 3 else (1 << 27) * 8
    out = {}
    for counter, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        rows = load(d, counter)
        c8 = mean(v for k, v in rows if "stream_copy_kernel<double>" in k) * 1024
        c16 = mean(v for k, v in rows if "stream_copy_kernel<HIP_vector_type<double, 2" in k or
                   ("stream_copy_kernel" in k and "double>" not in k.split("stream_copy_kernel")[1][:10])) * 1024
        skel = mean(v for k, v in rows if "traffic_skeleton" in k) * 1024
        caar = [(k, v * 1024) for k, v in rows if "caar_np" in k and "steps_kernel" not in k]
        steps = [(k, v * 1024) for k, v in rows if "caar_np" in k and "steps_kernel" in k]
        out[counter] = {
            "copy8_raw_bytes": c8, "copy16_raw_bytes": c16, "copy_true_bytes": copy_bytes,
            "factor_8B_lane": copy_bytes / c8 if c8 else None,
            "factor_16B_lane": copy_bytes / c16 if c16 else None,
            "skeleton_raw_bytes": skel,
            "caar_raw_bytes": mean(v for _, v in caar),
            "caar_kernel": caar[0][0] if caar else None,
            "caar_launches": len(caar),
            "steps_raw_bytes": mean(v for _, v in steps) if steps else None,
            "steps_kernel": steps[0][0] if steps else None,
            "steps_launches": len(steps),
        }
    f, w = out["FETCH_SIZE"], out["WRITE_SIZE"]
    res = {"counters": out}
    if f["factor_8B_lane"] and w["factor_8B_lane"]:
        res["caar_read_bytes_per_launch"] = f["caar_raw_bytes"] * f["factor_8B_lane"]
        res["caar_write_bytes_per_launch"] = w["caar_raw_bytes"] * w["factor_8B_lane"]
        res["hbm_bytes_per_launch"] = res["caar_read_bytes_per_launch"] + res["caar_write_bytes_per_launch"]
        res["skeleton_bytes_per_launch"] = f["skeleton_raw_bytes"] * f["factor_8B_lane"] + \
            w["skeleton_raw_bytes"] * w["factor_8B_lane"]
        if f["steps_raw_bytes"] is not None and w["steps_raw_bytes"] is not None:
            res["steps_read_bytes_per_launch"] = f["steps_raw_bytes"] * f["factor_8B_lane"]
            res["steps_write_bytes_per_launch"] = w["steps_raw_bytes"] * w["factor_8B_lane"]
            res["steps_hbm_bytes_per_launch"] = res["steps_read_bytes_per_launch"] + res["steps_write_bytes_per_launch"]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
