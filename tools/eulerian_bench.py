#!/usr/bin/env python3
"""Kernel time of the vertically-Lagrangian (rsplit > 0, the reference's path) and the
Eulerian (rsplit == 0) form on the same device-resident arrays, A/B in one process."""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import tinman_sandbox_amd as tsa  # noqa: E402


def time_ms(data, reps=20):
    for _ in range(10):  # (a variant's first launches load its code object)
        tsa.compute_and_apply_rhs(data)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        tsa.compute_and_apply_rhs(data)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for np_, nlev, E in ((4, 72, 10000), (4, 128, 12500), (8, 72, 20000)):
    data = tsa.TestData().init_data(E, np_, nlev, device="cuda")
    data.hvcoord.hybi = (np.arange(nlev + 1) / nlev) ** 2
    bytes_ = tsa.algorithmic_bytes(np_, nlev) * E
    out = []
    for rs in (1, 0, 1, 0):
        data.control.rsplit = rs
        ms = time_ms(data)
        out.append("rsplit=%d %.4f ms %.0f GB/s (%.1f%% of 8 TB/s)" % (rs, ms, bytes_ / ms / 1e6, bytes_ / ms / 8e7))
    print("np=%d nlev=%d E=%d: " % (np_, nlev, E) + " | ".join(out), flush=True)
    lib = tsa.library().lib
    data.control.rsplit = 0
    for v in range(lib.caar_num_variants(np_, nlev)):  # the Eulerian form of every launch shape
        lib.caar_select_variant(np_, nlev, v)
        ms = time_ms(data)
        print("    rsplit=0 variant %d  %.4f ms  %.0f GB/s  %s" % (v, ms, bytes_ / ms / 1e6,
                                                                   lib.caar_variant_info(np_, nlev, v).decode()), flush=True)
    lib.caar_select_variant(np_, nlev, 0)
    del data
    torch.cuda.empty_cache()
