#!/usr/bin/env python3
"""profiles/hbm_traffic.json from one round's counter passes: python tools/make_hbm_traffic.py r05

Reads profiles/<round>/pmc_traffic_np*_nlev*_e*.json (tools/profile_round.sh -> tools/pmc_parse.py: rocprofv3 --pmc FETCH_SIZE /
--pmc WRITE_SIZE in separate passes, each calibrated on the 8 B/lane stream copy of the same pass) and writes the table
bench.py's static_traffic() reads for the configurations it does not measure live."""
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
out = {}
for path in sorted(glob.glob(os.path.join(ROOT, "profiles", rnd, "pmc_traffic_np*_nlev*_e*.json"))):
    m = re.search(r"pmc_traffic_(np(\d+)_nlev(\d+)_e(\d+))\.json$", path)
    if not m or "steps" in path:
        continue
    tag, np_, nlev, e = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))
    j = json.load(open(path))
    balg = 8 * (21 * np_ * np_ * nlev + 2 * np_ * np_ * (nlev + 1) + 13 * np_ * np_) * e
    f, w = j["counters"]["FETCH_SIZE"], j["counters"]["WRITE_SIZE"]
    out[tag] = {
        "hbm_bytes_per_launch": j["hbm_bytes_per_launch"],
        "read_bytes": j["caar_read_bytes_per_launch"],
        "write_bytes": j["caar_write_bytes_per_launch"],
        "algorithmic_bytes_per_launch": balg,
        "ratio": j["hbm_bytes_per_launch"] / balg,
        "kernel": f["caar_kernel"],
        "source": "profiles/%s/%s" % (rnd, os.path.basename(path)),
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (KiB), each calibrated on the 8 B/lane stream "
                  "copy of the same run (known 1 GiB each way): FETCH x%.3f, WRITE x%.3f; tools/profile_round.sh (pmc_run.py + "
                  "pmc_parse.py)" % (f["factor_8B_lane"], w["factor_8B_lane"]),
        "note": "FETCH_SIZE / WRITE_SIZE count what leaves and enters the L2s; what the memory-side Infinity Cache then serves "
                "without HBM (the hybrid cache policy's accumulator blocks) is not subtracted",
    }
json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
for k, v in out.items():
    print(k, "x%.5f" % v["ratio"], v["kernel"][:80])
