#!/usr/bin/env python3
"""profiles/<round>/pmc_traffic_*.json (tools/profile_round.sh) -> profiles/hbm_traffic.json, the file bench.py reads
`roofline.traffic` from.     python tools/hbm_traffic_merge.py r03"""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rnd = sys.argv[1]
out = {}
for f in sorted(glob.glob(os.path.join(ROOT, "profiles", rnd, "pmc_traffic_np*_e*.json"))):
    m = re.search(r"pmc_traffic_np(\d+)_nlev(\d+)_e(\d+)\.json", f)
    np_, nlev, e = (int(x) for x in m.groups())
    j = json.load(open(f))
    pp = np_ * np_
    balg = 8 * (21 * pp * nlev + 2 * pp * (nlev + 1) + 13 * pp) * e
    fs, ws = j["counters"]["FETCH_SIZE"], j["counters"]["WRITE_SIZE"]
    out["np%d_nlev%d_e%d" % (np_, nlev, e)] = {
        "hbm_bytes_per_launch": j["hbm_bytes_per_launch"], "read_bytes": j["caar_read_bytes_per_launch"],
        "write_bytes": j["caar_write_bytes_per_launch"], "algorithmic_bytes_per_launch": balg,
        "ratio": j["hbm_bytes_per_launch"] / balg, "kernel": fs["caar_kernel"],
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (KiB), each calibrated on the 8 B/lane "
                  "stream copy of the same run (known 1 GiB each way): FETCH x%.3f, WRITE x%.3f; tools/profile_round.sh "
                  "(pmc_run.py + pmc_parse.py); raw JSON in profiles/%s/%s" % (fs["factor_8B_lane"], ws["factor_8B_lane"], rnd,
                                                                                os.path.basename(f)),
        "note": "FETCH_SIZE / WRITE_SIZE count what leaves and enters the L2s; what the memory-side Infinity Cache then "
                "serves without HBM (the hybrid cache policy's accumulator blocks) is not subtracted",
    }
json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
for k, v in out.items():
    print(k, "x%.5f" % v["ratio"])
