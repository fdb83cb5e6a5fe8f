#!/usr/bin/env python3
"""Hybrid cache policy (caar_set_cache_window): time per call of the default NP=4 kernels over
a sweep of the window, next to the all-streaming variant, A/B in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinman_sandbox_amd as tsa  # noqa: E402

lib = tsa.library().lib


def timed(data, n=40):
    for _ in range(5):
        tsa.compute_and_apply_rhs(data)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        tsa.compute_and_apply_rhs(data)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for nlev, E in ((72, 10000), (128, 12500)):
    data = tsa.TestData().init_data(E, 4, nlev, device="cuda")
    nt = 1  # variant 1 of every NP=4 table is the all-streaming twin of variant 0
    for rep in range(2):
        lib.caar_select_variant(4, nlev, nt)
        row = ["all-streaming %.4f" % timed(data)]
        lib.caar_select_variant(4, nlev, 0)
        for mb in (0, 64, 128, 160, 176, 192, 208, 224, 256, 320):
            lib.caar_set_cache_window(mb << 20)
            row.append("%dMB %.4f" % (mb, timed(data)))
        lib.caar_set_cache_window(224 << 20)
        print("nlev=%d E=%d  ms per call: " % (nlev, E) + " | ".join(row), flush=True)
    del data
    torch.cuda.empty_cache()
