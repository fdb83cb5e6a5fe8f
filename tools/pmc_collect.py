#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the counter passes of tools/profile_round.sh:
    python tools/pmc_collect.py gpurun_out/r02/prof r02
(copies the per-config results to profiles/<round>/ as well)."""
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src, rnd = sys.argv[1], sys.argv[2]
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
out = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_traffic_*.json"))):
    tag = re.search(r"pmc_traffic_(np\d+_nlev\d+_e\d+)\.json", f).group(1)
    np_, nlev, e = [int(x) for x in re.findall(r"\d+", tag)]
    j = json.load(open(f))
    balg = 8 * (21 * np_ * np_ * nlev + 2 * np_ * np_ * (nlev + 1) + 13 * np_ * np_) * e
    c = j["counters"]
    out[tag] = {
        "hbm_bytes_per_launch": j["hbm_bytes_per_launch"],
        "read_bytes": j["caar_read_bytes_per_launch"], "write_bytes": j["caar_write_bytes_per_launch"],
        "algorithmic_bytes_per_launch": balg, "ratio": j["hbm_bytes_per_launch"] / balg,
        "kernel": c["FETCH_SIZE"]["caar_kernel"],
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (KiB), each calibrated on the 8 B/lane "
                  "stream copy of the same run (known 1 GiB each way): FETCH x%.3f, WRITE x%.3f; tools/profile_round.sh "
                  "(pmc_run.py + pmc_parse.py); raw JSON in profiles/%s/pmc_traffic_%s.json" % (
                      c["FETCH_SIZE"]["factor_8B_lane"], c["WRITE_SIZE"]["factor_8B_lane"], rnd, tag),
        "note": "FETCH_SIZE / WRITE_SIZE count what leaves and enters the L2s; what the memory-side Infinity Cache then "
                "serves without HBM (the hybrid cache policy's accumulator blocks) is not subtracted",
    }
    shutil.copy(f, os.path.join(dst, os.path.basename(f)))
json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
for k, v in out.items():
    print("%-22s x%.5f of algorithmic  %s" % (k, v["ratio"], v["kernel"][:80]))
