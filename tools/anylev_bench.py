#!/usr/bin/env python3
"""Throughput of the NP=4 kernel with a run-time level count over a sweep of level counts
(specialised kernels where they exist, for comparison)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinman_sandbox_amd as tsa  # noqa: E402

lib = tsa.library().lib
for nlev in (20, 26, 32, 40, 50, 60, 64, 70, 72, 80, 90, 96, 100, 112, 128, 160, 200, 256):
    E = max(1000, int(10000 * 72 / nlev))
    data = tsa.TestData().init_data(E, 4, nlev, device="cuda")
    for _ in range(80):  # past the ramp of a fresh allocation (DESIGN.md section 5 "Cold start")
        tsa.compute_and_apply_rhs(data)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(40):
        tsa.compute_and_apply_rhs(data)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 40
    byts = tsa.algorithmic_bytes(4, nlev) * E
    print("nlev=%3d E=%6d  %.4f ms  %5.0f GB/s (%.1f%% of 8 TB/s)  %s" % (
        nlev, E, ms, byts / ms / 1e6, byts / ms / 8e7, lib.caar_kernel_name(4, nlev).decode()), flush=True)
    del data
    torch.cuda.empty_cache()
