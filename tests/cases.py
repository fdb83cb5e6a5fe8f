"""Seeded input cases shared by the fixture generator (tests/golden/make_golden.py)
and the parity tests.  Inputs are never stored: they are rebuilt here from
integer-hash pseudo-random numbers (splitmix64 on uint64: exactly reproducible on
every platform) or from the reference's closed-form initialiser; only the
reference's OUTPUTS live in tests/golden/*.npz.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import pyoracle as po  # noqa: E402  (tests may use the oracle)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
OUTPUT_NAMES = ("elem_state_dp3d", "elem_state_v", "elem_state_T",
                "elem_derived_eta_dot_dpdn", "elem_derived_omega_p",
                "elem_derived_phi", "elem_derived_vn0")


def splitmix64(idx, seed):
    """uint64 -> uint64, vectorised; pure integer arithmetic (wraps mod 2^64)."""
    with np.errstate(over="ignore"):
        z = idx.astype(np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(shape, seed, lo=0.0, hi=1.0):
    n = int(np.prod(shape))
    u = (splitmix64(np.arange(n, dtype=np.uint64), seed) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (lo + (hi - lo) * u).reshape(shape)


def hashed_arrays(np_, nlev, ne, seed, qsize_d=1, timelevels=3):
    """Physically plausible pseudo-random element arrays with a FULL (non-diagonal)
    D/Dinv, non-zero eta_dot_dpdn and every time level / Qdp slot populated, so
    that index or component mix-ups cannot cancel (the closed-form initialiser has
    diagonal D and symmetric fields)."""
    sh = po.array_shapes(np_, nlev, qsize_d, timelevels, ne)
    a = {}
    s = seed * 100
    D = uniform(sh["elem_D"], s + 1, -1.0, 1.0)
    D[..., 0, 0] += 2.0
    D[..., 1, 1] += 2.5
    a["elem_D"] = D
    det = D[..., 0, 0] * D[..., 1, 1] - D[..., 0, 1] * D[..., 1, 0]
    Dinv = np.empty_like(D)
    Dinv[..., 0, 0] = D[..., 1, 1] / det
    Dinv[..., 0, 1] = -D[..., 0, 1] / det
    Dinv[..., 1, 0] = -D[..., 1, 0] / det
    Dinv[..., 1, 1] = D[..., 0, 0] / det
    a["elem_Dinv"] = Dinv
    a["elem_fcor"] = uniform(sh["elem_fcor"], s + 2, -1.5e-4, 1.5e-4)
    a["elem_spheremp"] = uniform(sh["elem_spheremp"], s + 3, 0.1, 1.0)
    a["elem_metdet"] = uniform(sh["elem_metdet"], s + 4, 0.5, 2.0)
    a["elem_rmetdet"] = 1.0 / a["elem_metdet"]
    a["elem_state_dp3d"] = uniform(sh["elem_state_dp3d"], s + 5, 500.0, 1500.0)
    a["elem_state_v"] = uniform(sh["elem_state_v"], s + 6, -40.0, 40.0)
    a["elem_state_T"] = uniform(sh["elem_state_T"], s + 7, 200.0, 310.0)
    a["elem_state_phis"] = uniform(sh["elem_state_phis"], s + 8, 0.0, 3.0e4)
    a["elem_state_Qdp"] = uniform(sh["elem_state_Qdp"], s + 9, 0.0, 20.0)
    a["elem_derived_eta_dot_dpdn"] = uniform(sh["elem_derived_eta_dot_dpdn"], s + 10, -1.0, 1.0)
    a["elem_derived_omega_p"] = uniform(sh["elem_derived_omega_p"], s + 11, -1e-3, 1e-3)
    a["elem_derived_phi"] = uniform(sh["elem_derived_phi"], s + 12, 0.0, 1e5)
    a["elem_derived_pecnd"] = uniform(sh["elem_derived_pecnd"], s + 13, -50.0, 50.0)
    a["elem_derived_vn0"] = uniform(sh["elem_derived_vn0"], s + 14, -1e4, 1e4)
    return {k: np.ascontiguousarray(v) for k, v in a.items()}


def dvv_for(np_, kind="double"):
    O = po.Oracle()
    if np_ == 4 and kind == "double":
        return O.dvv_np4(False)
    if np_ == 4 and kind == "f32":
        return O.dvv_np4(True)
    return O.dvv_gll(np_)


# name -> dict(np, nlev, ne, init, dvv, overrides of default scalars)
CASES = {
    # the reference's own configuration (data_structures.cpp:117-163), double-literal Dvv
    "np4_nlev72_closed": dict(np=4, nlev=72, ne=3, init="closed", dvv="double", sc={}),
    # the Fortran driver's configuration: float32-rounded Dvv (main.F90:83-96)
    "np4_nlev72_closed_f32dvv": dict(np=4, nlev=72, ne=3, init="closed", dvv="f32", sc={}),
    # dry branch (P:128-139)
    "np4_nlev72_closed_dry": dict(np=4, nlev=72, ne=2, init="closed", dvv="double", sc=dict(qn0=-1)),
    # full metric tensors, permuted time levels, second Qdp slot, element sub-range,
    # non-unit dt2 / eta_ave_w
    "np4_nlev72_hashed": dict(np=4, nlev=72, ne=4, init="hashed", seed=1, dvv="double",
                              sc=dict(n0=2, np1=0, nm1=1, qn0=1, dt2=37.5, eta_ave_w=0.625,
                                      nets=1, nete=3)),
    # same with the horizontal-operator terms amplified (rrearth 1e-2 instead of 1.6e-7)
    # so that an error in any Dvv contraction is O(1) in the outputs
    "np4_nlev72_hashed_amplified": dict(np=4, nlev=72, ne=2, init="hashed", seed=2, dvv="double",
                                        sc=dict(n0=1, np1=2, nm1=0, qn0=0, dt2=0.01,
                                                eta_ave_w=0.5, rrearth=1e-2)),
    "np4_nlev128_closed": dict(np=4, nlev=128, ne=2, init="closed", dvv="double", sc={}),
    "np4_nlev128_hashed": dict(np=4, nlev=128, ne=2, init="hashed", seed=3, dvv="double",
                               sc=dict(n0=1, np1=2, nm1=0, qn0=1, dt2=12.0, eta_ave_w=0.75,
                                       rrearth=1e-3)),
    # NP=8: the reference has no derivative matrix for it; GLL matrix of this repo
    "np8_nlev72_closed": dict(np=8, nlev=72, ne=1, init="closed", dvv="gll", sc={}),
    "np8_nlev72_hashed": dict(np=8, nlev=72, ne=2, init="hashed", seed=4, dvv="gll",
                              sc=dict(n0=2, np1=1, nm1=0, qn0=0, dt2=5.0, eta_ave_w=0.25,
                                      rrearth=1e-3)),
}


def make_case(name):
    """-> (arrays dict, Dvv, scalars dict)"""
    c = CASES[name]
    if c["init"] == "closed":
        arrs = po.Oracle().init_arrays(c["np"], c["nlev"], 1, 3, c["ne"])
    else:
        arrs = hashed_arrays(c["np"], c["nlev"], c["ne"], c["seed"])
    sc = po.default_scalars(c["nlev"])
    sc.update(c["sc"])
    return arrs, dvv_for(c["np"], c["dvv"]), sc


def golden_path(name):
    return os.path.join(GOLDEN_DIR, name + ".npz")


def load_golden(name):
    with np.load(golden_path(name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def copy_arrays(arrs):
    return {k: v.copy() for k, v in arrs.items()}


def rel_err(got, want):
    """max |got-want| / max(|want|, tiny), elementwise; 0 where both are 0."""
    d = np.abs(got - want)
    den = np.maximum(np.abs(want), np.finfo(np.float64).tiny)
    return float(np.max(np.where(d == 0, 0.0, d / den))) if d.size else 0.0


def scaled_err(got, want):
    """max |got-want| / max|want|: error relative to the field's magnitude (used for
    accumulated diagnostics whose individual entries can cancel to ~0)."""
    m = float(np.max(np.abs(want)))
    return float(np.max(np.abs(got - want))) / (m if m > 0 else 1.0)
