#!/usr/bin/env python3
"""Instruction-issue profile of the CAAR kernels from the SQ counters (MI355X_MICROARCH.md profiling recipe: MFMA use,
LDS conflicts), beside the HBM-traffic passes of tools/pmc_run.py.

    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES \
              --kernel-trace --output-format csv -d out_a -- python3 tools/pmc_issue.py run
    rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT \
              SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d out_b -- python3 tools/pmc_issue.py run
    python3 tools/pmc_issue.py parse out_a out_b > profiles/rNN/pmc_issue.json

`run`: five single calls at NP=4 NLEV=72 (10 000 elements), NLEV=128 (12 500), NP=8 (20 000) and one 20-call step loop
each.  `parse`: per kernel, the counters per launch (averaged over the launches but the first) and per element-call, and
the ratios that say how busy the issue ports were.  Counter semantics are the profiler's (per-SE sums of per-SIMD events);
the per-element instruction counts are checked against the disassembly (180 MFMA per NP=4 NLEV=72 element: 10 contractions x
18 tiles)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CASES = ((4, 72, 10000), (4, 128, 12500), (8, 72, 20000))
NSTEPS = 20


def run():
    import torch
    import tinman_sandbox_amd as tsa
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev)
    for np_, nlev, E in CASES:
        data = tsa.TestData().init_data(E, np_, nlev, device=dev, place=tsa.placement("malloc"))
        for _ in range(5):
            tsa.compute_and_apply_rhs(data, st)
        tsa.compute_and_apply_rhs_steps(data, NSTEPS, True, st)
        torch.cuda.synchronize()
        del data
        torch.cuda.empty_cache()
    print("pmc_issue run done")


def read(d):
    """{kernel name: {counter: [value per dispatch, in dispatch order]}} and grid sizes"""
    out, grids = {}, {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            rows = list(csv.DictReader(f))
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            k = r["Kernel_Name"]
            if "caar_np" not in k:
                continue
            out.setdefault(k, {}).setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
            out[k][r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
            grids[k] = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
    return out, grids


def parse(dirs):
    merged, grids = {}, {}
    for d in dirs:
        o, g = read(d)
        grids.update(g)
        for k, cs in o.items():
            merged.setdefault(k, {}).update(cs)
    res = {}
    for k, cs in merged.items():
        steps = "steps_kernel" in k
        calls = NSTEPS if steps else 1
        elems = grids[k]
        row = {"workgroups": elems, "calls_per_launch": calls}
        for c, per_dispatch in cs.items():
            vals = [per_dispatch[i] for i in sorted(per_dispatch)]
            vals = vals[1:] if len(vals) > 1 else vals
            per_launch = sum(vals) / len(vals)
            row[c] = {"per_launch": per_launch, "per_element_call": per_launch / (elems * calls)}
        g = lambda n: row.get(n, {}).get("per_launch")  # noqa: E731
        if g("SQ_INSTS_VALU") and g("SQ_INSTS_MFMA") is not None:
            row["mfma_share_of_valu_instructions"] = g("SQ_INSTS_MFMA") / g("SQ_INSTS_VALU")
        if g("SQ_BUSY_CYCLES") and g("SQ_ACTIVE_INST_VALU") is not None:
            row["valu_active_over_busy_cycles"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_BUSY_CYCLES")
        if g("SQ_BUSY_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            row["mfma_busy_over_busy_cycles"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES")
        if g("SQ_ACTIVE_INST_LDS") and g("SQ_LDS_BANK_CONFLICT") is not None:
            row["lds_bank_conflict_over_lds_active_cycles"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_ACTIVE_INST_LDS")
        if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_INST_ANY") is not None:
            row["wave_cycles_waiting_for_any_instruction"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
        # Normalised to the hardware: SQ_BUSY_CYCLES is a per-shader-engine clock count (32 SEs on MI355X: sum / 32 = the
        # kernel's length in clocks; the r03 passes give 2.35 GHz that way), SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count
        # quad-cycles (4 clocks: one wave64 VALU instruction occupies its SIMD's issue port for one), SQ_VALU_MFMA_BUSY_CYCLES
        # clocks of a SIMD's matrix pipe (16 per v_mfma_f64_4x4x4).  1024 SIMDs.
        NSE, NSIMD = 32, 1024
        if g("SQ_BUSY_CYCLES"):
            clocks = g("SQ_BUSY_CYCLES") / NSE
            row["kernel_clocks"] = clocks
            if g("SQ_ACTIVE_INST_VALU") is not None:
                row["valu_issue_utilisation"] = 4.0 * g("SQ_ACTIVE_INST_VALU") / (NSIMD * clocks)
            if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
                row["mfma_pipe_utilisation"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (NSIMD * clocks)
            if g("SQ_WAVE_CYCLES") is not None:
                row["waves_per_simd"] = 4.0 * g("SQ_WAVE_CYCLES") / (NSIMD * clocks)
        res[k.split("(")[0].replace("void caar::", "")] = row
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "run":
        run()
    elif len(sys.argv) >= 3 and sys.argv[1] == "parse":
        parse(sys.argv[2:])
    else:
        sys.exit(__doc__)
