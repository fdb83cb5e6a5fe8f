#!/usr/bin/env python3
"""Workload for the HBM-traffic counter passes (run under rocprofv3 --pmc ...):
a few launches each of the 8 B/lane and 16 B/lane stream copies (KNOWN byte counts: the
calibration the MI355X guide asks for), of the traffic skeleton and of the CAAR kernel.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_fetch -- python3 tools/pmc_run.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_write -- python3 tools/pmc_run.py
    python3 tools/pmc_parse.py out_fetch out_write > profiles/...
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import tinman_sandbox_amd as tsa  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--np", type=int, default=4, dest="np_")
ap.add_argument("--nlev", type=int, default=72)
ap.add_argument("--elems", type=int, default=10000)
ap.add_argument("--rsplit", type=int, default=1, help="0: the Eulerian form (eta_dot_dpdn, vertical advection)")
ap.add_argument("--steps", type=int, default=0,
                help="also 3 launches of the driver loop (caar_launch_steps, this many calls per launch, time levels rotating): "
                     "the bytes the step-loop kernel really moves (tools/pmc_parse.py: steps_*)")
a = ap.parse_args()
L = tsa.library()
lib = L.lib
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
n_copy = 1 << 27  # 1 GiB each way
src = torch.ones(n_copy, dtype=torch.float64, device=dev)
dst = torch.empty_like(src)
data = tsa.TestData().init_data(a.elems, a.np_, a.nlev, device=dev, place=tsa.placement("malloc"))  # bytes, not rates
dims, ptrs, prm = data.arrays.dims(), data.arrays.pointers(), data.params()
torch.cuda.synchronize()
for lb in (8, 16):
    for _ in range(3):
        L.check(lib.caar_stream_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), n_copy, lb,
                                     C.c_void_p(st.cuda_stream)), "copy")
    torch.cuda.synchronize()
if a.np_ == 4:
    for _ in range(3):
        L.check(lib.caar_traffic_skeleton(C.byref(dims), C.byref(ptrs), C.byref(prm), 0, C.c_void_p(st.cuda_stream)), "skel")
    torch.cuda.synchronize()
    data = tsa.TestData().init_data(a.elems, a.np_, a.nlev, device=dev, place=tsa.placement("malloc"))
if a.rsplit == 0:
    import numpy as np
    data.hvcoord.hybi = (np.arange(a.nlev + 1) / a.nlev) ** 2
    data.control.rsplit = 0
for _ in range(5):
    tsa.compute_and_apply_rhs(data, st)
torch.cuda.synchronize()
if a.steps:
    data.control.dt2, data.constants.eta_ave_w = 1.0e-6, 0.0   # keeps the leap-frog steps finite (same as bench.py's leg)
    for _ in range(3):
        tsa.compute_and_apply_rhs_steps(data, a.steps, True, st)
    torch.cuda.synchronize()
print("pmc_run done: copy bytes each way = %d, caar B_alg per launch = %d" % (
    n_copy * 8, tsa.algorithmic_bytes(a.np_, a.nlev) * a.elems))
