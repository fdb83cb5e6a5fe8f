#!/usr/bin/env python3
"""Bandwidth of the Fortran-layout ingest / egress kernels (caar_layout_from_f90 / _to_f90):
bytes read + written per call over the HIP-event time, next to a plain device copy of the
same arrays."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinman_sandbox_amd as tsa  # noqa: E402
from tinman_sandbox_amd import f90_layout as fl  # noqa: E402


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for np_, nlev, E in ((4, 72, 10000), (4, 128, 12500), (8, 72, 20000)):
    data = tsa.TestData().init_data(E, np_, nlev, device="cuda")
    f90 = fl.F90Arrays(np_, nlev, E, device="cuda")
    all_bytes = sum(data.arrays[n].numel() for n in tsa.ARRAY_NAMES) * 8
    mut_bytes = sum(data.arrays[n].numel() for n in tsa.caar.MUTATED) * 8
    t_out_all = timed(lambda: fl.egress(data.arrays, f90, all_arrays=True))
    t_in = timed(lambda: fl.ingest(f90, data.arrays))
    t_out = timed(lambda: fl.egress(data.arrays, f90))

    def plain():
        for n in tsa.ARRAY_NAMES:
            f90.t[n].view(-1).copy_(data.arrays[n].view(-1))
    t_copy = timed(plain)
    gb = lambda byts, ms: 2 * byts / ms / 1e6
    print("np=%d nlev=%d E=%d: ingest all %.3f ms %.0f GB/s | egress all %.3f ms %.0f GB/s | egress mutated %.3f ms %.0f GB/s"
          " | torch copy of all %.3f ms %.0f GB/s" % (np_, nlev, E, t_in, gb(all_bytes, t_in), t_out_all, gb(all_bytes, t_out_all),
                                                      t_out, gb(mut_bytes, t_out), t_copy, gb(all_bytes, t_copy)), flush=True)
    del data, f90
    torch.cuda.empty_cache()
