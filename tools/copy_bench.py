#!/usr/bin/env python3
"""Device-copy ceiling on this box: every variant of caar_stream_copy_tuned and the naive 8 / 16 B
grid-stride copies, three buffer sizes, interleaved rounds (box drift hits all variants alike).
Bytes counted = read + written.  Log: profiles/r02/copy_bench.log"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import tinman_sandbox_amd as tsa  # noqa: E402

L = tsa.library()
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
sv = C.c_void_p(st.cuda_stream)


def timed(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / reps * 1e-3


nv = L.lib.caar_stream_copy_tuned_variants()
for log2n in (26, 27, 28):
    n = 1 << log2n
    src = torch.ones(n, dtype=torch.float64, device=dev)
    dst = torch.empty_like(src)
    sp, dp = C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr())
    best = {}
    for rnd in range(3):
        for v in list(range(nv)) + [-8, -16]:
            if v >= 0:
                t = timed(lambda: L.check(L.lib.caar_stream_copy_tuned(dp, sp, n, v, sv), "copy"))
            else:
                t = timed(lambda: L.check(L.lib.caar_stream_copy(dp, sp, n, -v, sv), "copy"))
            best.setdefault(v, []).append(2 * n * 8 / t / 1e9)
    print("buffer %d MiB each way" % (n * 8 >> 20))
    for v, g in best.items():
        name = L.lib.caar_stream_copy_tuned_info(v).decode() if v >= 0 else "naive grid-stride, %d B/lane" % -v
        print("  %-78s %s  GB/s" % (name, " ".join("%7.1f" % x for x in g)))
    del src, dst
    torch.cuda.empty_cache()
