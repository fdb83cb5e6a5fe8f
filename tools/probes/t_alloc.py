import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, ctypes as C, tinman_sandbox_amd as tsa
from tinman_sandbox_amd import caar as m
L = tsa.library()
torch.cuda.init(); torch.zeros(1, device="cuda")
for E in (2000, 10000, 100000):
    for rep in range(3):
        dims = m._CaarDims(4, 72, 1, 3, E)
        h, p = C.c_void_p(), m._CaarArrays()
        t0 = time.perf_counter()
        L.check(L.lib.caar_arrays_alloc(C.byref(h), C.byref(dims), 0, C.byref(p)), "alloc")
        t1 = time.perf_counter()
        L.lib.caar_arrays_free(h)
        t2 = time.perf_counter()
        print("E=%6d alloc %.3f s  free %.3f s" % (E, t1 - t0, t2 - t1), flush=True)
