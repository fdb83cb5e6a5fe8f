#!/usr/bin/env python3
"""The default NP=4 NLEV=72 kernel against its own traffic with no arithmetic, in several launch shapes, over a sweep of the
cache window: caar_traffic_skeleton variants 28-35 move the kernel's bytes with the kernel's hybrid cache policy.  Which
shape does the memory system serve fastest — is there a traffic pattern worth restructuring the kernel for?

    python tools/skeleton_hybrid_bench.py [--elems 10000]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinman_sandbox_amd as tsa  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--elems", type=int, default=10000)
a = ap.parse_args()
L = tsa.library()
lib = L.lib
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
data = tsa.TestData().init_data(a.elems, 4, 72, device=dev)
dims, ptrs, prm = data.arrays.dims(), data.arrays.pointers(), data.params()
balg = tsa.algorithmic_bytes(4, 72) * a.elems
lib.caar_set_adaptive_window(0)


def timed(fn, n=30):
    for _ in range(6):
        fn()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(n):
            fn()
        e1.record(st)
        torch.cuda.synchronize(dev)
        best = min(best, e0.elapsed_time(e1) / n)
    return best


names = {28: "3 waves x 6 tiles", 29: "6 x 3", 30: "18 one-wave workgroups per element", 31: "9 one-wave workgroups (2 tiles)",
         32: "9 two-wave workgroups", 33: "3 x 6, tile by tile", 34: "2 x 9", 35: "9 x 2"}
for mb in (224, 0, 192, 240):
    lib.caar_set_cache_window(mb << 20)
    ms = timed(lambda: tsa.compute_and_apply_rhs(data, st))
    print("window %3d MiB  kernel (default)                          %.4f ms  %.1f %% of peak" % (mb, ms, balg / ms / 8e7), flush=True)
    for v in sorted(names):
        ms = timed(lambda: L.check(lib.caar_traffic_skeleton(C.byref(dims), C.byref(ptrs), C.byref(prm), v, C.c_void_p(st.cuda_stream)), "skel"))
        print("window %3d MiB  skeleton %2d  %-36s %.4f ms  %.1f %% of peak" % (mb, v, names[v], ms, balg / ms / 8e7), flush=True)
lib.caar_set_cache_window(224 << 20)
lib.caar_set_adaptive_window(1)
