"""rsplit=0 kernel time of chosen variants: python tools/probes/eulerian_variants.py <np> <nlev> <elems> 0 14 15"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, tinman_sandbox_amd as tsa
np_, nlev, E = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
data = tsa.TestData().init_data(E, np_, nlev, device="cuda")
data.hvcoord.hybi = (np.arange(nlev + 1) / nlev) ** 2
data.control.rsplit = 0
lib = tsa.library().lib
B = tsa.algorithmic_bytes(np_, nlev) * E
def t(n):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): tsa.compute_and_apply_rhs(data)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
t(150)
for rep in range(2):
    for v in [int(x) for x in sys.argv[4:]]:
        lib.caar_select_variant(np_, nlev, v)
        t(20)
        ms = min(t(20), t(20))
        print("variant %2d  %.4f ms  %.1f %%  %s" % (v, ms, B / ms / 8e7, lib.caar_variant_info(np_, nlev, v).decode()[:70]), flush=True)
lib.caar_select_variant(np_, nlev, 0)
