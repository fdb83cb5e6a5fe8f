#!/usr/bin/env python3
"""Bandwidth of the stand-alone sphere operators over a whole element range
(caar_sphere_operator_range): bytes read + written over the HIP-event time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinman_sandbox_amd as tsa  # noqa: E402

for np_, nlev, E in ((4, 72, 10000), (4, 128, 12500), (8, 72, 20000)):
    data = tsa.TestData().init_data(E, np_, nlev, device="cuda")
    s = data.arrays["elem_state_T"][:, 0].contiguous()
    v = data.arrays["elem_state_v"][:, 0].contiguous()
    row = []
    for which, name, f in ((0, "gradient", s), (1, "divergence", v), (2, "vorticity", v)):
        for _ in range(3):
            tsa.sphere_operator_all(which, f, data)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            out = tsa.sphere_operator_all(which, f, data)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        byts = (f.numel() + out.numel()) * 8
        row.append("%s %.3f ms %.0f GB/s" % (name, ms, byts / ms / 1e6))
    print("np=%d nlev=%d E=%d: " % (np_, nlev, E) + " | ".join(row), flush=True)
    del data
    torch.cuda.empty_cache()
