#!/usr/bin/env python3
"""Prints, for every case of tests/cases.py the GPU can run, the error of the HIP path
against the CPU oracle: elementwise relative error and error scaled by the field's
max-abs, per output array; and, against an 80-bit evaluation of the same formulas
(oracle/np_oracle.py with numpy.longdouble), the rounding error of the reference (= the
oracle, bit-identical) next to the rounding error of the HIP path.
(Diagnostic, run by hand: `python tests/parity_report.py`; it lives under tests/ because it
uses the oracle, which only test code may; the assertions are in tests/test_parity_gpu.py.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # cases.py
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cases  # noqa: E402
from oracle import np_oracle  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
import tinman_sandbox_amd as tsa  # noqa: E402

O = po.Oracle()
for name, c in cases.CASES.items():
    if not tsa.library().lib.caar_supported(c["np"], c["nlev"]):
        continue
    arrs, Dvv, sc = cases.make_case(name)
    want = cases.copy_arrays(arrs)
    O.compute_and_apply_rhs(want, Dvv, sc)
    truth = np_oracle.compute_and_apply_rhs(arrs, Dvv, sc, dtype=np.longdouble)

    def vs_truth(x):
        return max(float(np.abs(x[n] - truth[n]).max() / max(float(np.abs(truth[n]).max()), 1e-300))
                   for n in cases.OUTPUT_NAMES)
    print("%-28s reference vs 80-bit evaluation: %.1e (worst field, scaled)" % (name, vs_truth(want)))
    nv = tsa.library().lib.caar_num_variants(c["np"], c["nlev"])
    for v in range(nv):
        tsa.library().lib.caar_select_variant(c["np"], c["nlev"], v)
        data = tsa.TestData.from_numpy(arrs, Dvv, sc, device="cuda")
        tsa.compute_and_apply_rhs(data)
        torch.cuda.synchronize()
        got = data.arrays.to_numpy()
        row = []
        for n in cases.OUTPUT_NAMES:
            row.append("%s rel %.1e scl %.1e" % (n.replace("elem_", "").replace("state_", "").replace("derived_", ""),
                                                  cases.rel_err(got[n], want[n]), cases.scaled_err(got[n], want[n])))
        print("%-28s v%d  HIP vs 80-bit %.1e | vs oracle: %s" % (name, v, vs_truth(got), " | ".join(row)))
    tsa.library().lib.caar_select_variant(c["np"], c["nlev"], 0)
