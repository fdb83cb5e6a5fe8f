"""Pins oracle/caar_oracle.c (the CPU parity checker) to the reference.

 1. the reference's own golden vectors Ttest/v1test/v2test (fortran/test_mod.F90:8-882,
    checked the way fortran/main.F90:241-274 checks them),
 2. committed outputs of the reference C++ path (cxx/pointers_only) and Fortran routine
    (fortran/routine_mod.F90) for every case in tests/cases.py (tests/golden/*.npz,
    produced by tests/golden/make_golden.py from oracle/_ref),
 3. when oracle/_ref is present (build container), the live reference library,
    operator by operator.
"""
import os

import numpy as np
import pytest

import cases
from oracle import pyoracle as po


def run_oracle(oracle, name):
    arrs, Dvv, sc = cases.make_case(name)
    out = cases.copy_arrays(arrs)
    oracle.compute_and_apply_rhs(out, Dvv, sc)
    return arrs, out, sc


@pytest.mark.parametrize("name", list(cases.CASES))
def test_oracle_matches_reference_cxx_bitwise(oracle, name):
    """Same operand order + no FMA contraction => bit-identical to g++ -O3 reference."""
    arrs, out, sc = run_oracle(oracle, name)
    gold = cases.load_golden(name)
    for n in cases.OUTPUT_NAMES:
        got = out[n][:, sc["np1"]] if n.startswith("elem_state_") else out[n]
        assert np.array_equal(got, gold[n]), n
    # nothing but np1 / the derived accumulators may change
    for n in po.ARRAY_NAMES:
        if n.startswith("elem_state_") and n in cases.OUTPUT_NAMES:
            for t in range(3):
                if t != sc["np1"]:
                    assert np.array_equal(out[n][:, t], arrs[n][:, t]), (n, t)
        elif n not in cases.OUTPUT_NAMES:
            assert np.array_equal(out[n], arrs[n]), n


@pytest.mark.parametrize("name", [n for n in cases.CASES if os.path.exists(cases.golden_path(n))])
def test_oracle_matches_reference_fortran(oracle, name):
    gold = cases.load_golden(name)
    if "f90_elem_state_T" not in gold:
        pytest.skip("no Fortran build of the reference for this configuration")
    c = cases.CASES[name]
    arrs, Dvv, sc = cases.make_case(name)
    sc["nets"], sc["nete"] = 0, None  # the Fortran fixture covers every element
    out = cases.copy_arrays(arrs)
    oracle.compute_and_apply_rhs(out, Dvv, sc)
    for n in cases.OUTPUT_NAMES:
        got = out[n][:, sc["np1"]] if n.startswith("elem_state_") else out[n]
        # flang -O2 and gcc agree to rounding; 1e-13 leaves room for association
        # differences (derivative_mod_base.F90:119,228 scale by (rmetdet*rrearth))
        assert cases.scaled_err(got, gold["f90_" + n]) <= 1e-13, (n, c)


def test_reference_golden_vectors_test_mod(oracle):
    """fortran/main.F90:241-274: element 1, time level np1, index i + np*(j-1) + np*np*(k-1),
    f32-rounded Dvv.  The reference accepts 'T diff 0, V diff 5.1e-13' (its own stdout,
    tests/golden/fortran_orig_stdout.txt)."""
    with np.load(os.path.join(cases.GOLDEN_DIR, "fortran_test_mod_vectors.npz")) as z:
        Tt, v1t, v2t = z["Ttest"], z["v1test"], z["v2test"]
    arrs, out, sc = run_oracle(oracle, "np4_nlev72_closed_f32dvv")
    # golden order: i fastest, then j, then k  <->  C++ [k][i][j]  => transpose in-level
    T = out["elem_state_T"][0, sc["np1"]].transpose(0, 2, 1).ravel()
    v1 = out["elem_state_v"][0, sc["np1"], ..., 0].transpose(0, 2, 1).ravel()
    v2 = out["elem_state_v"][0, sc["np1"], ..., 1].transpose(0, 2, 1).ravel()
    assert np.max(np.abs(T - Tt)) == 0.0
    assert np.max(np.abs(v1 - v1t)) <= 5.2e-13
    assert np.max(np.abs(v2 - v2t)) <= 5.2e-13
    assert cases.rel_err(v1, v1t) <= 1e-14 and cases.rel_err(v2, v2t) <= 1e-14


def test_reference_driver_norms(oracle):
    """The norms the reference drivers print (P:372-399, main.F90:168-194,278-304)."""
    txt = open(os.path.join(cases.GOLDEN_DIR, "fortran_orig_stdout.txt")).read().split()
    vals = [float(txt[i + 2]) for i, w in enumerate(txt) if w.startswith("||")]
    before, after = vals[:3], vals[3:6]
    arrs, Dvv, sc = cases.make_case("np4_nlev72_closed_f32dvv")
    assert np.allclose(oracle.state_norms(arrs, Dvv, sc), before, rtol=1e-15, atol=0)
    oracle.compute_and_apply_rhs(arrs, Dvv, sc)
    assert np.allclose(oracle.state_norms(arrs, Dvv, sc), after, rtol=2e-16, atol=0)


def test_idempotent_state_and_doubling_accumulators(oracle):
    """calling twice gives the same np1 state and doubles the increments."""
    arrs, Dvv, sc = cases.make_case("np4_nlev72_hashed")
    a1 = cases.copy_arrays(arrs)
    oracle.compute_and_apply_rhs(a1, Dvv, sc)
    a2 = cases.copy_arrays(a1)
    oracle.compute_and_apply_rhs(a2, Dvv, sc)
    for n in ("elem_state_v", "elem_state_T", "elem_state_dp3d", "elem_derived_phi"):
        assert np.array_equal(a1[n], a2[n])
    inc1 = a1["elem_derived_vn0"] - arrs["elem_derived_vn0"]
    inc2 = a2["elem_derived_vn0"] - a1["elem_derived_vn0"]
    assert cases.scaled_err(inc2, inc1) < 1e-12


def test_gll_matrix_reproduces_np4_literals(oracle):
    assert np.max(np.abs(oracle.dvv_gll(4) - oracle.dvv_np4())) < 2e-15
    D8 = oracle.dvv_gll(8)
    # derivative of a constant vanishes; exact for polynomials up to degree 7
    assert np.max(np.abs(D8.sum(axis=1))) < 1e-13


def _independent_gll_nodes(np_):
    """GLL nodes by a route neither generator uses: +-1 and the eigenvalue-based roots numpy finds
    for the monomial-basis polynomial d/dx P_N (both generators run Newton on Legendre recurrences)."""
    from numpy.polynomial import legendre as leg, polynomial as poly
    N = np_ - 1
    mono = leg.leg2poly([0.0] * N + [1.0])          # P_N in the monomial basis
    inner = np.sort(np.real(poly.polyroots(poly.polyder(mono))))
    return np.concatenate(([-1.0], inner, [1.0]))


@pytest.mark.parametrize("np_", [4, 8])
@pytest.mark.parametrize("which", ["oracle_c", "host_python"])
def test_derivative_matrix_pinned_without_its_generator(oracle, np_, which):
    """NP=8 has no reference literals (data_structures.cpp:152-162 hard-codes np=4),
    and both sides of the NP=8 parity tests use this repo's matrix — so the matrix itself is pinned
    by properties that DEFINE it, checked with nodes obtained independently of both generators:
      * Dvv[i][j] = l_j'(x_i) (the reference's storage, Dvv[i][j] = values[j*np+i]): exact
        differentiation of every monomial x^k, k = 0..np-1, at the GLL nodes;
      * antisymmetry under the node reflection x -> -x: D[i][j] = -D[N-i][N-j];
      * summation by parts with the GLL weights w_i = 2/(N(N+1)P_N(x_i)^2): W D + (W D)^T =
        diag(-1, 0, .., 0, 1) — holds only on the true GLL nodes (quadrature exact to degree 2N-1);
      * np=4: the same checks pass on the reference's literals, so the convention is the reference's."""
    import tinman_sandbox_amd as tsa
    D = oracle.dvv_gll(np_) if which == "oracle_c" else tsa.gll_derivative_matrix(np_)
    N = np_ - 1
    x = _independent_gll_nodes(np_)
    for k in range(np_):
        want = k * x ** (k - 1) if k > 0 else np.zeros(np_)
        assert np.max(np.abs(D @ x ** k - want)) < 2e-13 * max(1, k * k), k
    assert np.max(np.abs(D + D[::-1, ::-1])) < 1e-13
    PN = np.polynomial.legendre.Legendre.basis(N)(x)
    W = np.diag(2.0 / (N * (N + 1) * PN ** 2))
    B = np.zeros((np_, np_))
    B[0, 0], B[N, N] = -1.0, 1.0
    assert np.max(np.abs(W @ D + (W @ D).T - B)) < 1e-13
    if np_ == 4:  # the reference's own literals satisfy the same definitions
        R = oracle.dvv_np4(False)
        for k in range(4):
            want = k * x ** (k - 1) if k > 0 else np.zeros(4)
            assert np.max(np.abs(R @ x ** k - want)) < 1e-14
        assert np.max(np.abs(R - D)) < 2e-15


needs_ref = pytest.mark.skipif(not po.have_reference(4, 72),
                               reason="oracle/_ref not built (needs /root/reference)")


@needs_ref
@pytest.mark.parametrize("np_,nlev", [(4, 72), (8, 72)])
def test_live_reference_operators(oracle, np_, nlev):
    R = po.Reference(np_, nlev)
    arrs = cases.hashed_arrays(np_, nlev, 2, seed=11)
    Dvv = cases.dvv_for(np_, "double" if np_ == 4 else "gll")
    rr = 0.37
    for ie in range(2):
        s = cases.uniform((np_, np_), 77 + ie, -3, 5)
        v = cases.uniform((np_, np_, 2), 99 + ie, -3, 5)
        g = oracle.gradient_sphere(s, Dvv, arrs["elem_Dinv"][ie], rr)
        d = oracle.divergence_sphere(v, Dvv, arrs["elem_Dinv"][ie], arrs["elem_metdet"][ie],
                                     arrs["elem_rmetdet"][ie], rr)
        w = oracle.vorticity_sphere(v, Dvv, arrs["elem_D"][ie], arrs["elem_rmetdet"][ie], rr)
        assert np.array_equal(g, R.sphere_operator(0, s, arrs, ie, rr, Dvv))
        assert np.array_equal(d, R.sphere_operator(1, v, arrs, ie, rr, Dvv))
        assert np.array_equal(w, R.sphere_operator(2, v, arrs, ie, rr, Dvv))


@needs_ref
def test_live_reference_init_and_norm(oracle):
    R = po.Reference(4, 72)
    arrs_r, Dvv_r, sc_r = R.init_data(5)
    arrs_o = oracle.init_arrays(4, 72, 1, 3, 5)
    for n in po.ARRAY_NAMES:
        assert np.array_equal(arrs_r[n], arrs_o[n]), n
    assert np.array_equal(Dvv_r, oracle.dvv_np4(False))
    sc = po.default_scalars(72)
    for k in ("n0", "np1", "nm1", "qn0", "dt2", "rrearth", "eta_ave_w", "Rwater_vapor", "Rgas", "kappa", "ps0"):
        assert sc[k] == sc_r[k], k
    assert np.array_equal(sc["hyai"], sc_r["hyai"])
    f = arrs_o["elem_state_v"][0, 1]
    assert oracle.compute_norm(f) == R.compute_norm(f)


@needs_ref
@pytest.mark.parametrize("levels", [(0, 1, 0), (1, 1, 0), (0, 1, 1), (2, 2, 2)])
def test_live_reference_aliased_time_levels(oracle, levels):
    """Coinciding time-level indices (HOMME's Runge-Kutta stages): the oracle follows the
    reference's statement order (all n0 reads before the update loop, P:236-257), so it
    has to stay bit-identical when n0, np1 and nm1 alias."""
    R = po.Reference(4, 72)
    arrs = cases.hashed_arrays(4, 72, 2, seed=140)
    Dvv = cases.dvv_for(4)
    sc = po.default_scalars(72)
    sc.update(n0=levels[0], np1=levels[1], nm1=levels[2], qn0=1, dt2=0.25, eta_ave_w=0.5)
    a, b = cases.copy_arrays(arrs), cases.copy_arrays(arrs)
    oracle.compute_and_apply_rhs(a, Dvv, sc)
    R.compute_and_apply_rhs(b, Dvv, sc)
    for n in po.ARRAY_NAMES:
        assert np.array_equal(a[n], b[n]), n


def eulerian_case(np_=4, nlev=72, ne=2, seed=150):
    arrs = cases.hashed_arrays(np_, nlev, ne, seed=seed)
    Dvv = cases.dvv_for(np_)
    sc = po.default_scalars(nlev)
    eta = np.arange(nlev + 1) / nlev
    sc.update(qn0=1, dt2=0.25, eta_ave_w=0.5, rsplit=0, hybi=eta ** 2)   # B(eta): 0 at the top, 1 at the surface
    return arrs, Dvv, sc


def test_eulerian_branch_against_numpy_restatement(oracle):
    """rsplit == 0 (PARITY UNPINNED: the reference neither builds nor tests this branch, see
    oracle/caar_oracle.h).  What can be checked: the C restatement against an independent
    numpy evaluation of routine_extracted.F90:224-262,515-517 and CaarFunctor.hpp:505-547,
    fed with divdp recovered from the (pinned) vertically-Lagrangian run of the same inputs."""
    arrs, Dvv, sc = eulerian_case()
    lag, eul = cases.copy_arrays(arrs), cases.copy_arrays(arrs)
    oracle.compute_and_apply_rhs(lag, Dvv, dict(sc, rsplit=1))
    oracle.compute_and_apply_rhs(eul, Dvv, sc)
    n0, np1, nm1, dt2, w = sc["n0"], sc["np1"], sc["nm1"], sc["dt2"], sc["eta_ave_w"]
    sph = arrs["elem_spheremp"][:, None]
    dp_nm1 = arrs["elem_state_dp3d"][:, nm1]
    divdp = (dp_nm1 - lag["elem_state_dp3d"][:, np1] / sph) / dt2            # P:254 inverted
    S = divdp.sum(axis=1, keepdims=True)
    eta_dot = np.zeros_like(arrs["elem_derived_eta_dot_dpdn"])
    eta_dot[:, 1:-1] = sc["hybi"][1:-1, None, None] * S - np.cumsum(divdp, axis=1)[:, :-1]
    scale = np.abs(S).max()
    got_eta = (eul["elem_derived_eta_dot_dpdn"] - arrs["elem_derived_eta_dot_dpdn"]) / w
    assert np.abs(got_eta - eta_dot).max() <= 1e-9 * scale
    assert np.array_equal(lag["elem_derived_eta_dot_dpdn"], arrs["elem_derived_eta_dot_dpdn"] + 0.0)
    # dp3d: X:515-517
    want_dp = sph * (dp_nm1 - dt2 * (divdp + eta_dot[:, 1:] - eta_dot[:, :-1]))
    assert np.abs(eul["elem_state_dp3d"][:, np1] - want_dp).max() <= 1e-9 * np.abs(want_dp).max()
    # vertical advection of T and v (CaarFunctor.hpp:505-547) through the np1 difference
    rdp = 1.0 / arrs["elem_state_dp3d"][:, n0]

    def vadv(f, ed, r):  # f: [ne][nlev][np][np](...) ; ed broadcastable
        d = np.zeros_like(f)
        up = 0.5 * r[:, :-1] * ed[:, 1:-1] * (f[:, 1:] - f[:, :-1])
        d[:, :-1] += up                                                       # facp term of level k
        d[:, 1:] += 0.5 * r[:, 1:] * ed[:, 1:-1] * (f[:, 1:] - f[:, :-1])     # facm term of level k+1
        return d
    T = arrs["elem_state_T"][:, n0]
    want_T = lag["elem_state_T"][:, np1] - sph * dt2 * vadv(T, eta_dot, rdp)
    assert np.abs(eul["elem_state_T"][:, np1] - want_T).max() <= 1e-9 * np.abs(want_T).max()
    v = arrs["elem_state_v"][:, n0]
    want_v = lag["elem_state_v"][:, np1] - (sph * dt2)[..., None] * vadv(v, eta_dot[..., None], rdp[..., None])
    assert np.abs(eul["elem_state_v"][:, np1] - want_v).max() <= 1e-9 * np.abs(want_v).max()
    # everything the vertical terms do not enter is bit-identical to the Lagrangian run
    for n in ("elem_derived_phi", "elem_derived_omega_p", "elem_derived_vn0"):
        assert np.array_equal(eul[n], lag[n]), n


def test_eulerian_branch_conserves_column_mass(oracle):
    """The interface fluxes telescope (eta_dot = 0 at the top and at the surface, X:253-254):
    the column sum of dp3d(np1)/spheremp is the one of the vertically-Lagrangian update."""
    arrs, Dvv, sc = eulerian_case(seed=151)
    lag, eul = cases.copy_arrays(arrs), cases.copy_arrays(arrs)
    oracle.compute_and_apply_rhs(lag, Dvv, dict(sc, rsplit=1))
    oracle.compute_and_apply_rhs(eul, Dvv, sc)
    a = eul["elem_state_dp3d"][:, sc["np1"]].sum(axis=1)
    b = lag["elem_state_dp3d"][:, sc["np1"]].sum(axis=1)
    assert np.abs(a - b).max() <= 1e-11 * np.abs(b).max()
    assert not np.array_equal(eul["elem_state_dp3d"], lag["elem_state_dp3d"])


@pytest.mark.parametrize("name", list(cases.CASES))
def test_numpy_restatement_agrees_with_the_c_oracle(oracle, name):
    """oracle/np_oracle.py: the path restated a second time, by different code (whole-column
    cumulative sums, einsum contractions), must land on the C oracle — which is bit-identical
    to the reference — to rounding.  In 80-bit arithmetic it is the yardstick for the
    rounding error of the reference itself (<= 1e-13 of the field magnitude on every case)."""
    from oracle import np_oracle
    arrs, out, sc = run_oracle(oracle, name)
    Dvv = cases.make_case(name)[1]
    for dtype, tol in ((np.float64, 2e-13), (np.longdouble, 1e-13)):
        want = np_oracle.compute_and_apply_rhs(arrs, Dvv, sc, dtype=dtype)
        for n in np_oracle.MUTATED:
            err = float(np.abs(out[n] - want[n]).max() / max(np.abs(want[n]).max(), 1e-300))
            assert err <= tol, (name, dtype.__name__, n, err)


def test_numpy_restatement_eulerian_and_aliasing(oracle):
    from oracle import np_oracle
    arrs, Dvv, sc = eulerian_case(seed=152)
    for extra in (dict(), dict(n0=1, np1=1, nm1=0), dict(qn0=-1, nets=1, nete=2)):
        s = dict(sc, **extra)
        got = cases.copy_arrays(arrs)
        oracle.compute_and_apply_rhs(got, Dvv, s)
        want = np_oracle.compute_and_apply_rhs(arrs, Dvv, s, dtype=np.longdouble)
        for n in np_oracle.MUTATED:
            err = float(np.abs(got[n] - want[n]).max() / max(np.abs(want[n]).max(), 1e-300))
            assert err <= 1e-13, (sorted(extra), n, err)
