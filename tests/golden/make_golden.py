#!/usr/bin/env python3
"""Regenerates tests/golden/*.  Runs ONLY in the build container, where
/root/reference exists and oracle/Makefile has built oracle/_ref/ from it; the
committed fixtures are what travels (the reference itself never does).

Produces
  <case>.npz                      outputs of the reference C++ path
                                  (cxx/pointers_only, via oracle/_ref/libref_caar_*.so)
                                  for every case of tests/cases.py: time level np1 of
                                  state_{v,T,dp3d} + the four mutated derived arrays;
                                  key prefix "f90_" = the same from the reference
                                  Fortran routine (oracle/_ref/fortran_driver) where
                                  that build applies (NP=4, NLEV=72, default constants)
  f90_native_np4_nlev72_hashed.npz  the reference Fortran routine's arrays after one call, written
                                  in Fortran's own array-element order (write(u) elem(ie)%state%v ...)
                                  by oracle/_ref/fortran_driver: keys "c_<name>" = C++-layout result,
                                  "f_<name>" = the same array as Fortran stores it
  fortran_test_mod_vectors.npz    Ttest / v1test / v2test, the reference's own golden
                                  vectors, parsed as numbers from
                                  compute_and_apply_rhs_test/fortran/test_mod.F90:8-882
  fortran_orig_stdout.txt         stdout of the reference's unmodified Fortran driver
                                  (oracle/_ref/fortran_orig: golden self-check + norms)
"""
import os
import re
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

REF_F90 = "/root/reference/compute_and_apply_rhs_test/fortran"


def outputs_of(arrs, sc):
    o = {}
    for n in cases.OUTPUT_NAMES:
        o[n] = arrs[n][:, sc["np1"]].copy() if n.startswith("elem_state_") else arrs[n].copy()
    return o


def parse_test_mod():
    txt = open(os.path.join(REF_F90, "test_mod.F90")).read()
    out = {}
    for name in ("Ttest", "v1test", "v2test"):
        m = re.search(r"::\s*" + name + r"\(np\*np\*nlev\)\s*=\s*\(/(.*?)/\)", txt, re.S)
        nums = re.findall(r"[-+]?\d+\.\d*(?:[DdEe][-+]?\d+)?", m.group(1))
        vals = np.array([float(x.replace("D", "E").replace("d", "e")) for x in nums])
        assert vals.size == 4 * 4 * 72, (name, vals.size)
        out[name] = vals
    return out


def main():
    po.build(ref=True)
    for name, c in cases.CASES.items():
        arrs, Dvv, sc = cases.make_case(name)
        R = po.Reference(c["np"], c["nlev"])
        a = cases.copy_arrays(arrs)
        R.compute_and_apply_rhs(a, Dvv, sc)
        out = outputs_of(a, sc)
        # the reference must not touch anything else
        for n in po.ARRAY_NAMES:
            if n in cases.OUTPUT_NAMES and n.startswith("elem_state_"):
                for t in range(3):
                    if t != sc["np1"]:
                        assert np.array_equal(a[n][:, t], arrs[n][:, t])
            elif n not in cases.OUTPUT_NAMES:
                assert np.array_equal(a[n], arrs[n])
        fortran_ok = (c["np"], c["nlev"]) == (4, 72) and "rrearth" not in c["sc"]
        if fortran_ok:
            # the Fortran routine has no nets/nete sub-range in this driver: run all elements
            scf = dict(sc)
            scf["nets"], scf["nete"] = 0, None
            fo = po.run_fortran_driver(arrs, Dvv, scf)
            for n in cases.OUTPUT_NAMES:
                out["f90_" + n] = fo[n][:, sc["np1"]].copy() if n.startswith("elem_state_") else fo[n]
        np.savez_compressed(cases.golden_path(name), **out)
        print("wrote", name, {k: v.shape for k, v in out.items() if not k.startswith("f90_")},
              "+f90" if fortran_ok else "")

    # Fortran-native array order (pin for the layout kernels): the hashed case, first 2 elements
    arrs, Dvv, sc = cases.make_case("np4_nlev72_hashed")
    scf = dict(sc)
    scf["nets"], scf["nete"] = 0, None
    native = po.run_fortran_driver(arrs, Dvv, scf, native=True)
    np.savez_compressed(os.path.join(HERE, "f90_native_np4_nlev72_hashed.npz"),
                        **{k: v[:2] for k, v in native.items()})
    print("wrote f90_native_np4_nlev72_hashed.npz", {k: v[:2].shape for k, v in native.items()})

    np.savez_compressed(os.path.join(HERE, "fortran_test_mod_vectors.npz"), **parse_test_mod())
    res = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "fortran_orig")],
                         check=True, capture_output=True, text=True)
    keep = [l for l in res.stdout.splitlines() if not l.lower().lstrip().startswith(("time", "raw time"))]
    open(os.path.join(HERE, "fortran_orig_stdout.txt"), "w").write("\n".join(keep) + "\n")
    print("wrote fortran_test_mod_vectors.npz, fortran_orig_stdout.txt")


if __name__ == "__main__":
    main()
