#!/usr/bin/env python3
"""caar_launch_steps / caar_run_steps: nsteps calls with rotating time levels as ONE launch (caar_np4_steps_kernel) against
the same calls as single launches, per element count and cache policy of the step-loop kernel.

    python tools/steps_bench.py [--nlev 72] [--nsteps 20] [--elems 64,256,1024,4096,10000]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinman_sandbox_amd as tsa  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--np", type=int, default=4, dest="np_")
ap.add_argument("--nlev", type=int, default=72)
ap.add_argument("--nsteps", type=int, default=20)
ap.add_argument("--elems", default="64,256,1024,4096,10000")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
lib = tsa.library().lib
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
variants = [v for v in range(lib.caar_num_variants(a.np_, a.nlev)) if lib.caar_has_fused_steps(a.np_, a.nlev, v)]
balg = tsa.algorithmic_bytes(a.np_, a.nlev)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize(dev)
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            fn()
        e1.record(st)
        torch.cuda.synchronize(dev)
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


print("NP=%d NLEV=%d, %d calls per run_steps, time levels rotating; ms per CALL (algorithmic TB/s)" % (a.np_, a.nlev, a.nsteps))
for E in [int(x) for x in a.elems.split(",")]:
    data = tsa.TestData().init_data(E, a.np_, a.nlev, device=dev)
    data.constants.eta_ave_w = 0.0   # keeps the accumulators finite over thousands of calls (timing only)
    data.control.dt2 = 1e-6
    for v in variants:
        lib.caar_select_variant(a.np_, a.nlev, v)
        row = []
        for fused in (0, 1):
            lib.caar_set_fused_steps(fused)
            ms = timed(lambda: tsa.compute_and_apply_rhs_steps(data, a.nsteps, True, st), a.reps) / a.nsteps
            row.append(ms)
        print("E=%6d  variant %2d  single launches %.4f ms (%.2f)   fused %.4f ms (%.2f)   x%.3f   %s" % (
            E, v, row[0], balg * E / row[0] / 1e9, row[1], balg * E / row[1] / 1e9, row[0] / row[1],
            lib.caar_variant_info(a.np_, a.nlev, v).decode()[:60]), flush=True)
    lib.caar_set_fused_steps(1)
    lib.caar_select_variant(a.np_, a.nlev, 0)
    del data
