"""Second, independent CPU restatement of compute_and_apply_rhs — vectorised numpy, any
float dtype.  TEST INFRASTRUCTURE ONLY (imported by tests/ only).

Written from the mathematical statement of the path, not from
oracle/caar_oracle.c: whole-column cumulative sums and einsum contractions instead of the
reference's loops, so its rounding differs from the reference's in the last bits.  Two uses:
  * dtype=float64: cross-check of the C oracle by different code (tests/test_oracle.py);
  * dtype=numpy.longdouble (x87 80-bit, 64-bit mantissa): a higher-precision evaluation of
    the same formulas, against which the rounding error of the reference itself and of the
    HIP kernels can be compared (tests/test_parity_gpu.py, tests/parity_report.py).
Reference lines: P = cxx/pointers_only/compute_and_apply_rhs.cpp, S = sphere_operators.cpp,
X = fortran/routine_extracted.F90, K = cxx/level_vectorized_ppscan/CaarFunctor.hpp.
"""
import numpy as np

MUTATED = ("elem_state_dp3d", "elem_state_v", "elem_state_T", "elem_derived_eta_dot_dpdn",
           "elem_derived_omega_p", "elem_derived_phi", "elem_derived_vn0")


def gradient_sphere(s, Dvv, Dinv, rrearth):
    """S:9-48.  s [..., np, np] -> [..., np, np, 2]; Dinv [ne, np, np, 2, 2] broadcast over levels."""
    v1 = rrearth * np.einsum("il,...ij->...lj", Dvv, s)      # d/da
    v2 = rrearth * np.einsum("il,...ji->...jl", Dvv, s)      # d/db
    return np.stack([Dinv[..., 0, 0] * v1 + Dinv[..., 1, 0] * v2,
                     Dinv[..., 0, 1] * v1 + Dinv[..., 1, 1] * v2], axis=-1)


def divergence_sphere(v, Dvv, Dinv, metdet, rmetdet, rrearth):
    """S:50-89.  v [..., np, np, 2] -> [..., np, np]."""
    gv0 = metdet * (Dinv[..., 0, 0] * v[..., 0] + Dinv[..., 0, 1] * v[..., 1])
    gv1 = metdet * (Dinv[..., 1, 0] * v[..., 0] + Dinv[..., 1, 1] * v[..., 1])
    dudx = np.einsum("ki,...kj->...ij", Dvv, gv0)
    dvdy = np.einsum("kj,...ik->...ij", Dvv, gv1)
    return (dudx + dvdy) * rmetdet * rrearth


def vorticity_sphere(v, Dvv, D, rmetdet, rrearth):
    """S:91-129."""
    vco0 = D[..., 0, 0] * v[..., 0] + D[..., 1, 0] * v[..., 1]
    vco1 = D[..., 0, 1] * v[..., 0] + D[..., 1, 1] * v[..., 1]
    dvdx = np.einsum("ki,...kj->...ij", Dvv, vco1)
    dudy = np.einsum("kj,...ik->...ij", Dvv, vco0)
    return (dvdx - dudy) * rmetdet * rrearth


def compute_and_apply_rhs(arrs, Dvv, sc, dtype=np.float64):
    """Returns a dict with the seven mutated arrays (full shape, dtype `dtype`); `arrs` is not
    modified.  sc: the flat scalar dict of oracle/pyoracle.py (optionally rsplit=0 + hybi)."""
    A = {k: np.asarray(v, dtype=dtype) for k, v in arrs.items()}
    out = {k: A[k].copy() for k in MUTATED}
    ne = A["elem_fcor"].shape[0]
    nlev = A["elem_state_dp3d"].shape[2]
    e0, e1 = sc["nets"], (ne if sc.get("nete") is None else sc["nete"])
    if e1 <= e0:
        return out
    sl = slice(e0, e1)
    f = lambda x: dtype(x)  # noqa: E731
    n0, np1, nm1, qn0 = sc["n0"], sc["np1"], sc["nm1"], sc["qn0"]
    dt2, w, rr = f(sc["dt2"]), f(sc["eta_ave_w"]), f(sc["rrearth"])
    Rgas, kappa = f(sc["Rgas"]), f(sc["kappa"])
    half = f(0.5)
    Dvv = np.asarray(Dvv, dtype=dtype)
    lev = lambda x: x[:, None]  # noqa: E731  (broadcast an [ne, np, np(...)] field over levels)
    Dinv, Dm = lev(A["elem_Dinv"][sl]), lev(A["elem_D"][sl])
    metdet, rmetdet = lev(A["elem_metdet"][sl]), lev(A["elem_rmetdet"][sl])
    fcor, sph, phis = lev(A["elem_fcor"][sl]), lev(A["elem_spheremp"][sl]), A["elem_state_phis"][sl]

    dp = A["elem_state_dp3d"][sl, n0]
    v = A["elem_state_v"][sl, n0]
    T = A["elem_state_T"][sl, n0]
    u1, u2 = v[..., 0], v[..., 1]
    # a2: p = hyai0*ps0 + sum_{l<k} dp + dp/2                                      P:78-97
    cum = np.cumsum(dp, axis=1)
    p = f(sc["hyai"][0]) * f(sc["ps0"]) + (cum - dp) + half * dp
    grad_p = gradient_sphere(p, Dvv, Dinv, rr)                                   # P:103
    vgrad_p = u1 * grad_p[..., 0] + u2 * grad_p[..., 1]                          # P:111
    vdp = v * dp[..., None]                                                      # P:114-115
    out["elem_derived_vn0"][sl] = A["elem_derived_vn0"][sl] + w * vdp            # P:117-118
    divdp = divergence_sphere(vdp, Dvv, Dinv, metdet, rmetdet, rr)               # P:121
    vort = vorticity_sphere(v, Dvv, Dm, rmetdet, rr)                             # P:122
    if qn0 == -1:                                                                # P:128-139
        Tv = T
    else:                                                                        # P:141-155
        Qt = A["elem_state_Qdp"][sl, 0, qn0] / dp
        Tv = T * (f(1.0) + (f(sc["Rwater_vapor"]) / Rgas - f(1.0)) * Qt)
    # a8: phi = phis + sum_{l>k} Rgas Tv dp/p + Rgas Tv dp/(2p)                     P:280-312
    ht = Rgas * Tv * dp / p
    below = np.cumsum(ht[:, ::-1], axis=1)[:, ::-1] - ht
    phi = phis[:, None] + below + half * ht
    out["elem_derived_phi"][sl] = phi
    # a9: omega_p = (vgrad_p - sum_{l<k} divdp - divdp/2) / p                       P:314-352
    csum = np.cumsum(divdp, axis=1)
    omega = (vgrad_p - (csum - divdp) - half * divdp) / p
    out["elem_derived_omega_p"][sl] = A["elem_derived_omega_p"][sl] + w * omega    # P:173

    eta_dot = np.zeros((e1 - e0, nlev + 1) + dp.shape[2:], dtype=dtype)
    T_vadv = np.zeros_like(T)
    v_vadv = np.zeros_like(v)
    if int(sc.get("rsplit", 1)) == 0:                                            # X:224-262
        hybi = np.asarray(sc["hybi"], dtype=dtype)
        eta_dot[:, 1:-1] = hybi[1:-1, None, None] * csum[:, -1:] - csum[:, :-1]
        hr = half / dp
        dT = T[:, 1:] - T[:, :-1]                                                # K:505-547
        dv = v[:, 1:] - v[:, :-1]
        e_in = eta_dot[:, 1:-1]
        T_vadv[:, :-1] += hr[:, :-1] * e_in * dT
        T_vadv[:, 1:] += hr[:, 1:] * e_in * dT
        v_vadv[:, :-1] += (hr[:, :-1] * e_in)[..., None] * dv
        v_vadv[:, 1:] += (hr[:, 1:] * e_in)[..., None] * dv
    out["elem_derived_eta_dot_dpdn"][sl] = A["elem_derived_eta_dot_dpdn"][sl] + w * eta_dot  # P:172,181

    # a11 tendencies                                                              P:187-233
    Ephi = half * (u1 * u1 + u2 * u2) + phi + A["elem_derived_pecnd"][sl]
    gT = gradient_sphere(T, Dvv, Dinv, rr)
    vgrad_T = u1 * gT[..., 0] + u2 * gT[..., 1]
    gE = gradient_sphere(Ephi, Dvv, Dinv, rr)
    gpterm = Tv / p
    vt1 = -v_vadv[..., 0] + u2 * (fcor + vort) - gE[..., 0] - Rgas * gpterm * grad_p[..., 0]
    vt2 = -v_vadv[..., 1] - u1 * (fcor + vort) - gE[..., 1] - Rgas * gpterm * grad_p[..., 1]
    tt = -T_vadv - vgrad_T + kappa * Tv * omega
    # a12 update                                                                  P:236-257, X:515-517
    vm = A["elem_state_v"][sl, nm1]
    out["elem_state_v"][sl, np1] = np.stack([sph * (vm[..., 0] + dt2 * vt1), sph * (vm[..., 1] + dt2 * vt2)], axis=-1)
    out["elem_state_T"][sl, np1] = sph * (A["elem_state_T"][sl, nm1] + dt2 * tt)
    out["elem_state_dp3d"][sl, np1] = sph * (A["elem_state_dp3d"][sl, nm1] -
                                             dt2 * (divdp + eta_dot[:, 1:] - eta_dot[:, :-1]))
    return out


# ---- sphere operators next to the CAAR path: a second, independent statement -------------------------------
# Written from HOMME's Fortran formulas (derivative_mod: divergence_sphere_wk, laplace_sphere_wk,
# curl_sphere_wk_testcov, gradient_sphere_wk_testcov, vlaplace_sphere_wk_contra / _cartesian) directly in this
# repository's index convention (field[a][b] == Fortran (a+1, b+1), tensors [a][b][r][c] == Fortran
# (., ., r+1, c+1), Dvv[i][j] == Fortran Dvv(i+1, j+1)) with einsum contractions — NOT through the accessor
# macros of sphere_ops_oracle.c, so an index slip there does not repeat here.  Test infrastructure only.
def ops_gradient_sphere(s, Dvv, Dinv, rr):
    v1 = np.einsum("il,ij->lj", Dvv, s) * rr
    v2 = np.einsum("il,ji->jl", Dvv, s) * rr
    return np.stack([Dinv[..., 0, 0] * v1 + Dinv[..., 1, 0] * v2, Dinv[..., 0, 1] * v1 + Dinv[..., 1, 1] * v2], axis=-1)


def ops_divergence_sphere(v, Dvv, Dinv, metdet, rr):
    gv0 = metdet * (Dinv[..., 0, 0] * v[..., 0] + Dinv[..., 0, 1] * v[..., 1])
    gv1 = metdet * (Dinv[..., 1, 0] * v[..., 0] + Dinv[..., 1, 1] * v[..., 1])
    return (np.einsum("il,ij->lj", Dvv, gv0) + np.einsum("il,ji->jl", Dvv, gv1)) / metdet * rr


def ops_vorticity_sphere(v, Dvv, D, metdet, rr):
    vc0 = D[..., 0, 0] * v[..., 0] + D[..., 1, 0] * v[..., 1]
    vc1 = D[..., 0, 1] * v[..., 0] + D[..., 1, 1] * v[..., 1]
    return (np.einsum("il,ij->lj", Dvv, vc1) - np.einsum("il,ji->jl", Dvv, vc0)) / metdet * rr


def ops_divergence_sphere_wk(v, Dvv, Dinv, spheremp, rr):
    t0 = spheremp * (Dinv[..., 0, 0] * v[..., 0] + Dinv[..., 0, 1] * v[..., 1])
    t1 = spheremp * (Dinv[..., 1, 0] * v[..., 0] + Dinv[..., 1, 1] * v[..., 1])
    return -(np.einsum("mj,jn->mn", Dvv, t0) + np.einsum("nj,mj->mn", Dvv, t1)) * rr


def ops_laplace_tensor(s, Dvv, Dinv, spheremp, tensorVisc, rr):
    g = ops_gradient_sphere(s, Dvv, Dinv, rr)
    if tensorVisc is not None:
        g = np.stack([tensorVisc[..., 0, 0] * g[..., 0] + tensorVisc[..., 0, 1] * g[..., 1],
                      tensorVisc[..., 1, 0] * g[..., 0] + tensorVisc[..., 1, 1] * g[..., 1]], axis=-1)
    return ops_divergence_sphere_wk(g, Dvv, Dinv, spheremp, rr)


def _cov_to_sphere(D, c0, c1, rr):
    return np.stack([D[..., 0, 0] * c0 + D[..., 0, 1] * c1, D[..., 1, 0] * c0 + D[..., 1, 1] * c1], axis=-1) * rr


def ops_curl_sphere_wk_testcov(s, Dvv, D, mp, rr):
    ms = mp * s
    return _cov_to_sphere(D, -np.einsum("nj,mj->mn", Dvv, ms), np.einsum("mj,jn->mn", Dvv, ms), rr)


def ops_grad_sphere_wk_testcov(s, Dvv, D, mp, metinv, metdet, rr):
    ms = mp * s
    A = np.einsum("mj,jn->mn", Dvv, ms)
    B = np.einsum("nj,mj->mn", Dvv, ms)
    c0 = -(metinv[..., 0, 0] * metdet * A + metinv[..., 1, 0] * metdet * B)
    c1 = -(metinv[..., 0, 1] * metdet * A + metinv[..., 1, 1] * metdet * B)
    return _cov_to_sphere(D, c0, c1, rr)


def ops_vlaplace_sphere_wk_contra(v, Dvv, D, Dinv, mp, spheremp, metinv, metdet, nu_ratio, rr):
    div = ops_divergence_sphere(v, Dvv, Dinv, metdet, rr) * nu_ratio
    vort = ops_vorticity_sphere(v, Dvv, D, metdet, rr)
    return (2.0 * spheremp[..., None] * v * rr * rr + ops_grad_sphere_wk_testcov(div, Dvv, D, mp, metinv, metdet, rr)
            - ops_curl_sphere_wk_testcov(vort, Dvv, D, mp, rr))


def ops_vlaplace_sphere_wk_cartesian(v, Dvv, Dinv, spheremp, tensorVisc, s2c, rr, undamp_rr=True):
    lap = [ops_laplace_tensor(s2c[..., k, 0] * v[..., 0] + s2c[..., k, 1] * v[..., 1], Dvv, Dinv, spheremp, tensorVisc, rr)
           for k in range(3)]
    out = np.stack([sum(s2c[..., k, h] * lap[k] for k in range(3)) for h in range(2)], axis=-1)
    if undamp_rr:
        out = out + 2.0 * spheremp[..., None] * v * rr * rr
    return out
