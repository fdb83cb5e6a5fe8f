#!/usr/bin/env python3
"""Traffic skeletons (CAAR's bytes and addressing, no arithmetic) in their launch-shape / cache-policy
variants, against the default kernel and the best tuned copy, interleaved rounds in one process.
    python tools/skeleton_bench.py [--nlev 72] [--elems 10000] [--ids 0,8,15,16,...]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import tinman_sandbox_amd as tsa  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nlev", type=int, default=72)
ap.add_argument("--elems", type=int, default=10000)
ap.add_argument("--ids", type=str, default="8,13,15,16,17,18,19,20")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
L = tsa.library()
lib = L.lib
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
sv = C.c_void_p(st.cuda_stream)
data = tsa.TestData().init_data(a.elems, 4, a.nlev, device=dev)
dims, ptrs, prm = data.arrays.dims(), data.arrays.pointers(), data.params()
balg = tsa.algorithmic_bytes(4, a.nlev) * a.elems


def timed(fn):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(a.reps):
        fn()
    e1.record(st)
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / a.reps * 1e-3


n = 1 << 27
src = torch.ones(n, dtype=torch.float64, device=dev)
dst = torch.empty_like(src)
rows = {}
for rnd in range(a.rounds):
    lib.caar_set_cache_window(0)
    rows.setdefault("kernel (default variant, all streaming)", []).append(
        balg / timed(lambda: tsa.compute_and_apply_rhs(data, st)) / 1e9)
    lib.caar_set_cache_window(224 << 20)
    rows.setdefault("kernel (default variant, hybrid window)", []).append(
        balg / timed(lambda: tsa.compute_and_apply_rhs(data, st)) / 1e9)
    for i in [int(x) for x in a.ids.split(",")]:
        t = timed(lambda: L.check(lib.caar_traffic_skeleton(C.byref(dims), C.byref(ptrs), C.byref(prm), i, sv), "skel"))
        rows.setdefault("skeleton %d" % i, []).append(balg / t / 1e9)
    for v in (0, 15):
        t = timed(lambda: L.check(lib.caar_stream_copy_tuned(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), n, v, sv), "copy"))
        rows.setdefault("copy: " + lib.caar_stream_copy_tuned_info(v).decode(), []).append(2 * n * 8 / t / 1e9)
print("NP=4 NLEV=%d elems=%d, GB/s per round" % (a.nlev, a.elems))
for k, g in rows.items():
    print("  %-84s %s" % (k, " ".join("%7.1f" % x for x in g)))
