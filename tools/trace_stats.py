#!/usr/bin/env python3
"""rocprofv3 --kernel-trace output -> one row per (kernel, grid, workgroup): calls, average / min / max
duration, VGPRs, LDS, scratch.  `rocprofv3 --stats` aggregates by kernel NAME only; bench.py launches the same
kernel on two grids (10 000 elements: headline; 12 500: one GPU's share of configs[2]), which this separates.

    python tools/trace_stats.py <rocprof output dir> [name filter] > profiles/rNN/bench_kernel_stats_by_grid.csv"""
import csv
import glob
import os
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--window=")]
d = args[0]
flt = args[1] if len(args) > 1 else "caar::"
# --window=<bench.json>: also report the timed region of bench.py (roofline.timed_dispatches of its JSON line)
window = None
for a in sys.argv[1:]:
    if a.startswith("--window="):
        import json
        j = json.load(open(a.split("=", 1)[1]))
        window = (j["config"]["kernel"], j["roofline"]["elements_per_launch"], j["roofline"]["timed_dispatches"], j["roofline"]["kernel_ms"])
rows = {}
ordered = {}
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if flt not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"], int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]), int(r["VGPR_Count"]),
               int(r["Accum_VGPR_Count"]), int(r["LDS_Block_Size"]), int(r["Scratch_Size"]))
        rows.setdefault(key, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        ordered.setdefault(key, []).append((int(r["Dispatch_Id"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
w = csv.writer(sys.stdout)
w.writerow(["Kernel_Name", "Grid_Size_X", "Workgroup_Size_X", "Workgroups", "VGPR_Count", "Accum_VGPR_Count", "LDS_Block_Size",
            "Scratch_Size", "Calls", "AverageNs", "MinNs", "MaxNs", "AverageNs_without_first"])
for k, t in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    rest = t[1:] if len(t) > 1 else t
    w.writerow([k[0], k[1], k[2], k[1] // k[2], k[3], k[4], k[5], k[6], len(t), "%.1f" % (sum(t) / len(t)), min(t), max(t),
                "%.1f" % (sum(rest) / len(rest))])
if window:
    name, elems, (a, b), kms = window
    for k, t in ordered.items():
        if name in k[0] and k[1] // k[2] == elems:
            t = [x for _, x in sorted(t)][a:b]
            w.writerow(["TIMED REGION of bench.py (dispatches %d..%d of the kernel above on %d workgroups; bench.py kernel_ms %.4f)" % (
                a, b - 1, elems, kms), k[1], k[2], elems, k[3], k[4], k[5], k[6], len(t), "%.1f" % (sum(t) / len(t)), min(t), max(t), ""])
