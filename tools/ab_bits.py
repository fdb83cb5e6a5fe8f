#!/usr/bin/env python3
"""Bit-level fingerprint of what ONE build of the library computes, for an A/B of two builds that must not differ in a
single bit (pure data-movement changes: DPP scans instead of ds_bpermute, another transpose, another park stride):

    CAAR_LIBRARY_PATH=$PWD/tinman_sandbox_amd/csrc/libcaar_hip_old.so python tools/ab_bits.py > old.txt
    python tools/ab_bits.py > new.txt && diff old.txt new.txt

One line per (configuration, variant, mode): sha256 over every array the path mutates.  Inputs: the reference's closed-form
arrays and tests/cases.py's hashed arrays, plus the awkward ones for a scan rewrite — velocities that are exactly zero and
negative zeros (x + 0.0 is not the identity on -0.0), run-time level counts with dead rows, both vertical coordinates, the
step loops with rotating and aliased time levels."""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
import tinman_sandbox_amd as tsa  # noqa: E402
from oracle import pyoracle as po  # noqa: E402  (default scalars / closed-form arrays only: a tool, not the product)

lib = tsa.library().lib
dev = torch.device("cuda", 0)


def digest(data):
    h = hashlib.sha256()
    for n in tsa.caar.MUTATED:
        h.update(data.arrays[n].detach().cpu().numpy().tobytes())
    return h.hexdigest()[:24]


def inputs(np_, nlev, ne, kind, seed):
    if kind == "closed":
        return po.Oracle().init_arrays(np_, nlev, 1, 3, ne)
    a = cases.hashed_arrays(np_, nlev, ne, seed)
    if kind == "zeros":      # u = v = 0 on part of the points, -0.0 on others: divdp and vorticity hit signed zeros
        v = a["elem_state_v"]
        v[:, :, ::2] = 0.0
        v[:, :, 1::4] = -0.0
        a["elem_derived_vn0"][:, ::3] = -0.0
        a["elem_derived_omega_p"][:, ::2] = -0.0
    return a


def main():
    only_np = int(os.environ.get("AB_BITS_NP", "0"))
    configs = [(4, 72), (4, 128), (4, 80), (4, 60), (4, 26), (4, 96), (4, 50), (4, 100), (4, 200), (8, 72)]
    for np_, nlev in configs:
        if only_np and np_ != only_np:
            continue
        if not lib.caar_supported(np_, nlev):
            continue
        ne = 5 if np_ == 4 else 3
        dvv = cases.dvv_for(np_, "double" if np_ == 4 else "gll")
        for kind in ("closed", "hashed", "zeros"):
            for rsplit in (1, 0):
                for qn0 in (0, -1):
                    if (rsplit == 0 or qn0 == -1) and kind == "closed":
                        continue
                    sc = po.default_scalars(nlev)
                    sc.update(dict(qn0=qn0, dt2=0.37, eta_ave_w=0.625, rsplit=rsplit, n0=1, np1=2, nm1=0))
                    if kind != "closed":
                        sc["rrearth"] = 1e-3
                    if rsplit == 0:
                        sc["hybi"] = (np.arange(nlev + 1) / nlev) ** 2
                    arrs = inputs(np_, nlev, ne, kind, 7 + nlev)
                    for v in range(lib.caar_num_variants(np_, nlev)):
                        lib.caar_select_variant(np_, nlev, v)
                        d = tsa.TestData.from_numpy(cases.copy_arrays(arrs), dvv, sc, device=dev)
                        try:
                            for _ in range(3):
                                tsa.compute_and_apply_rhs(d)
                                d.update_time_levels()
                        except tsa.caar.CaarError:   # a form this build does not hold (rsplit == 0 beyond 128 levels)
                            print("np%d nlev%d %s rsplit%d qn0=%d variant %d: refused" % (np_, nlev, kind, rsplit, qn0, v))
                            continue
                        torch.cuda.synchronize()
                        print("np%d nlev%d %s rsplit%d qn0=%d variant %d single x3: %s" % (np_, nlev, kind, rsplit, qn0, v, digest(d)))
                        if rsplit == 1 and lib.caar_has_fused_steps(np_, nlev, v):
                            for rotate, alias in ((True, False), (False, False), (True, True)):
                                d = tsa.TestData.from_numpy(cases.copy_arrays(arrs), dvv, sc, device=dev)
                                if alias:
                                    d.control.nm1 = d.control.n0
                                d.control.dt2 = 1e-3
                                tsa.compute_and_apply_rhs_steps(d, 7, rotate)
                                torch.cuda.synchronize()
                                print("np%d nlev%d %s qn0=%d variant %d steps x7 rotate=%d alias=%d: %s" % (
                                    np_, nlev, kind, qn0, v, rotate, alias, digest(d)))
                    lib.caar_select_variant(np_, nlev, 0)
    # full-size: the launch shapes with two workgroups per CU, XCD-chunked mapping, cache window
    for np_, nlev, E in ((4, 72, 3000), (4, 128, 2000), (8, 72, 1500)):
        if only_np and np_ != only_np:
            continue
        d = tsa.TestData().init_data(E, np_, nlev, device=dev)
        for _ in range(2):
            tsa.compute_and_apply_rhs(d)
            d.update_time_levels()
        d.control.dt2, d.constants.eta_ave_w = 1e-6, 0.0
        tsa.compute_and_apply_rhs_steps(d, 5, True)
        torch.cuda.synchronize()
        print("np%d nlev%d closed E=%d single x2 + steps x5: %s" % (np_, nlev, E, digest(d)))


if __name__ == "__main__":
    main()
