"""Placement dependence, part 2: (a) does a plain copy alternate too when its buffers are freed and reallocated?
(b) two CAAR data sets alive at once: measured alternately, do they keep their own rate?"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, tinman_sandbox_amd as tsa
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
L = tsa.library()
sv = C.c_void_p(st.cuda_stream)
def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        fn()
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
n = 1 << 27
for trial in range(6):
    src = torch.ones(n, dtype=torch.float64, device=dev)
    dst = torch.empty_like(src)
    f = lambda: L.check(L.lib.caar_stream_copy_tuned(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), n, 15, sv), "c")
    timed(f, 30)
    r = [2 * n * 8 / timed(f, 10) / 1e6 for _ in range(3)]
    print("copy trial %d: %s GB/s  src 0x%x dst 0x%x" % (trial, " ".join("%.0f" % x for x in r), src.data_ptr(), dst.data_ptr()), flush=True)
    del src, dst
    torch.cuda.empty_cache()
balg = tsa.algorithmic_bytes(4, 72) * 10000
sets = [tsa.TestData().init_data(10000, 4, 72, device=dev) for _ in range(4)]
for rnd in range(3):
    out = []
    for d in sets:
        g = lambda: tsa.compute_and_apply_rhs(d, st)
        timed(g, 60)
        out.append(balg / timed(g, 20) / 8e7)
    print("4 data sets alive, round %d: %s %% of peak" % (rnd, " ".join("%.1f" % x for x in out)), flush=True)
print("state_v bases: " + " ".join("0x%x" % d.arrays["elem_state_v"].data_ptr() for d in sets))
