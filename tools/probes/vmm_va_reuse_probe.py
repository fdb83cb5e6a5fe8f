"""Allocate / release / allocate placed arenas of different sizes and check the results against the golden outputs each time:
shows the ROCm 7.2 defect worked around in csrc/caar_alloc.hip (a virtual range that is freed and reserved again keeps its old
GPU translations) when run against a build that frees its ranges, and nothing with the committed build."""
import sys, os, gc
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, ctypes as C
import cases
from oracle import pyoracle as po
import tinman_sandbox_amd as tsa
def run(np_, nlev, E, gold_name):
    data = tsa.TestData().init_data(E, np_, nlev, device="cuda")
    ar = data.arrays.arena
    ptrs = [C.cast(getattr(ar.ptrs, f), C.c_void_p).value for f, _ in tsa.caar._CaarArrays._fields_]
    same = all(data.arrays[n].data_ptr() == p for n, p in zip(tsa.ARRAY_NAMES, ptrs))
    tsa.compute_and_apply_rhs(data)
    torch.cuda.synchronize()
    gold = cases.load_golden(gold_name)
    ng = gold["elem_derived_phi"].shape[0]
    sc = po.default_scalars(nlev)
    errs = {}
    for n in cases.OUTPUT_NAMES:
        g = data.arrays[n][:ng].cpu().numpy()
        g = g[:, sc["np1"]] if n.startswith("elem_state_") else g
        errs[n[5:]] = cases.scaled_err(g, gold[n])
    print(np_, nlev, E, "spread", ar.spread(), "tensors view the arena:", same, {k: "%.1e" % v for k, v in errs.items()}, flush=True)
for rep in range(2):
    run(4, 72, 10000, "np4_nlev72_closed")
    run(4, 128, 12500, "np4_nlev128_closed")
    run(8, 72, 20000, "np8_nlev72_closed")
    run(8, 72, 2000, "np8_nlev72_closed")
