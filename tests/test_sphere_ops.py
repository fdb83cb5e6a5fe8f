"""Sphere operators next to the CAAR path: divergence_sphere_wk, laplace_simple,
laplace_tensor, curl/grad_sphere_wk_testcov, vlaplace_sphere_wk_contra/_cartesian, gradient/
divergence_sphere_update (reference: cxx/level_vectorized_ppscan/SphereOperators.hpp:271-993).

PARITY UNPINNED: the reference defines these only as Kokkos device functions it cannot build here, never
calls or tests them and holds no output of them.  This file is what pins the ORACLE of them instead
(CPU tests, first half):
  * the index mapping of oracle/sphere_ops_oracle.c is validated on the three PINNED operators
    (pointers_only gradient/divergence/vorticity, bit-identical to the reference's C++);
  * defining identities (weak divergence = negative adjoint of the pinned gradient, compositions);
  * an independent numpy statement from HOMME's Fortran formulas (oracle/np_oracle.py).
GPU tests (second half): the HIP operators (caar_sphere_operator_ex) against that oracle to 1e-12.
"""
import numpy as np
import pytest

import cases
from oracle import np_oracle as npo
from oracle import pyoracle as po

NPS = (4, 8)
RR = 0.37  # rrearth of order one, so that every term matters


def geometry(np_, ne, seed):
    """Random, well-conditioned per-element geometry incl. the arrays only these operators read."""
    a = cases.hashed_arrays(np_, 4, ne, seed)
    g = {"D": a["elem_D"], "Dinv": a["elem_Dinv"], "metdet": a["elem_metdet"], "rmetdet": a["elem_rmetdet"],
         "spheremp": a["elem_spheremp"]}
    g["mp"] = cases.uniform((ne, np_, np_), seed * 100 + 41, 0.05, 0.6)
    mi = cases.uniform((ne, np_, np_, 2, 2), seed * 100 + 42, -0.4, 0.4)
    mi[..., 0, 0] += 1.5
    mi[..., 1, 1] += 1.2
    g["metinv"] = 0.5 * (mi + mi.swapaxes(-1, -2))  # symmetric like a metric
    tv = cases.uniform((ne, np_, np_, 2, 2), seed * 100 + 43, -0.3, 0.3)
    tv[..., 0, 0] += 1.0
    tv[..., 1, 1] += 0.8
    g["tensorVisc"] = tv
    g["vec_sph2cart"] = cases.uniform((ne, np_, np_, 3, 2), seed * 100 + 44, -1.0, 1.0)
    return {k: np.ascontiguousarray(v) for k, v in g.items()}


def elem(g, ie):
    return {k: v[ie] for k, v in g.items()}


def fields(np_, seed):
    return cases.uniform((np_, np_), seed, -3, 5), cases.uniform((np_, np_, 2), seed + 1, -3, 5)


def close(a, b, tol=2e-13):
    return float(np.max(np.abs(a - b))) <= tol * max(1.0, float(np.max(np.abs(b))))


# ------------------------------------------------------------------------------------------- CPU: the oracle
@pytest.mark.parametrize("np_", NPS)
def test_index_mapping_reproduces_the_pinned_operators(oracle, np_):
    """The K: (ppscan) forms of gradient / divergence / vorticity, restated through the accessor macros every
    new operator uses, equal the pinned pointers_only operators: the macros map K:'s indices correctly."""
    Dvv = cases.dvv_for(np_, "double" if np_ == 4 else "gll")
    g = geometry(np_, 2, 5)
    for ie in range(2):
        s, v = fields(np_, 300 + ie)
        e = elem(g, ie)
        pinned_g = oracle.gradient_sphere(s, Dvv, e["Dinv"], RR)
        assert np.array_equal(po.sphere_op(oracle, "gradient_sphere", s, Dvv, e, RR), pinned_g)
        # K: multiplies by (1/metdet * rrearth), pointers_only by rmetdet and then rrearth: rounding only
        pinned_d = oracle.divergence_sphere(v, Dvv, e["Dinv"], e["metdet"], 1.0 / e["metdet"], RR)
        assert close(po.sphere_op(oracle, "divergence_sphere", v, Dvv, e, RR), pinned_d, 1e-15)
        pinned_w = oracle.vorticity_sphere(v, Dvv, e["D"], 1.0 / e["metdet"], RR)
        assert close(po.sphere_op(oracle, "vorticity_sphere_vector", v, Dvv, e, RR), pinned_w, 1e-15)


@pytest.mark.parametrize("np_", NPS)
def test_weak_divergence_is_the_negative_adjoint_of_the_pinned_gradient(oracle, np_):
    """sum_pts phi * div_wk(v) == - sum_pts spheremp * grad(phi) . v for every nodal phi: the definition of the
    weak divergence (integration by parts against the GLL basis), with the PINNED gradient_sphere."""
    Dvv = cases.dvv_for(np_, "double" if np_ == 4 else "gll")
    g = geometry(np_, 1, 6)
    e = elem(g, 0)
    _, v = fields(np_, 310)
    dw = po.sphere_op(oracle, "divergence_sphere_wk", v, Dvv, e, RR)
    for a in range(np_):
        for b in range(np_):
            phi = np.zeros((np_, np_))
            phi[a, b] = 1.0
            gphi = oracle.gradient_sphere(phi, Dvv, e["Dinv"], RR)
            rhs = -np.sum(e["spheremp"] * (gphi[..., 0] * v[..., 0] + gphi[..., 1] * v[..., 1]))
            assert abs(dw[a, b] - rhs) <= 1e-12 * max(1.0, abs(rhs)), (a, b)


@pytest.mark.parametrize("np_", NPS)
def test_compositions(oracle, np_):
    Dvv = cases.dvv_for(np_, "double" if np_ == 4 else "gll")
    g = geometry(np_, 1, 7)
    e = elem(g, 0)
    s, v = fields(np_, 320)
    grad = oracle.gradient_sphere(s, Dvv, e["Dinv"], RR)  # pinned
    lap = po.sphere_op(oracle, "laplace_simple", s, Dvv, e, RR)
    assert np.array_equal(lap, po.sphere_op(oracle, "divergence_sphere_wk", grad, Dvv, e, RR))
    ident = dict(e)
    ident["tensorVisc"] = np.broadcast_to(np.eye(2), (np_, np_, 2, 2)).copy()
    assert np.array_equal(po.sphere_op(oracle, "laplace_tensor", s, Dvv, ident, RR), lap)
    # K:600-637: the in-place form gives what the out-of-place one gives, and leaves nothing of the input behind
    assert np.array_equal(po.sphere_op(oracle, "laplace_tensor_replace", s, Dvv, e, RR),
                          po.sphere_op(oracle, "laplace_tensor", s, Dvv, e, RR))
    # a constant field has no gradient, hence no Laplacian — with the derivative matrix in HOMME's orientation
    # Dvv(i, l) = l_i'(x_l).  (The reference's standalone drivers fill Dvv the other way round,
    # data_structures.cpp:152-162 / main.F90:83-96: there the "gradient" of a constant is not zero, which is
    # what the pinned operators and golden vectors reproduce; the formulas are the same either way.)
    assert np.max(np.abs(po.sphere_op(oracle, "laplace_tensor", np.full((np_, np_), 3.25), Dvv.T.copy(), e, RR))) < 1e-11
    # update forms
    acc = cases.uniform((np_, np_, 2), 77, -1, 1)
    assert np.array_equal(po.sphere_op(oracle, "gradient_sphere_update", s, Dvv, e, RR, out=acc), acc + grad)
    accd = cases.uniform((np_, np_), 78, -1, 1)
    d = po.sphere_op(oracle, "divergence_sphere", v, Dvv, e, RR)
    got = po.sphere_op(oracle, "divergence_sphere_update", v, Dvv, e, RR, out=accd, alpha=0.75, beta=-1.5)
    assert np.array_equal(got, accd * -1.5 + 0.75 * d)
    # vlaplace_contra = 2 spheremp v rr^2 + grad_wk(nu div v) - curl_wk(vort v)
    div = po.sphere_op(oracle, "divergence_sphere", v, Dvv, e, RR) * 2.5
    vort = po.sphere_op(oracle, "vorticity_sphere_vector", v, Dvv, e, RR)
    want = 2.0 * e["spheremp"][..., None] * v * RR * RR
    want = want + (po.sphere_op(oracle, "grad_sphere_wk_testcov", div, Dvv, e, RR)
                   - po.sphere_op(oracle, "curl_sphere_wk_testcov", vort, Dvv, e, RR))
    assert close(po.sphere_op(oracle, "vlaplace_sphere_wk_contra", v, Dvv, e, RR, nu_ratio=2.5), want, 1e-15)


@pytest.mark.parametrize("np_", NPS)
def test_against_independent_numpy_statement(oracle, np_):
    Dvv = cases.dvv_for(np_, "double" if np_ == 4 else "gll")
    g = geometry(np_, 2, 8)
    for ie in range(2):
        e = elem(g, ie)
        s, v = fields(np_, 330 + ie)
        O = lambda name, x, **kw: po.sphere_op(oracle, name, x, Dvv, e, RR, **kw)  # noqa: E731
        assert close(O("gradient_sphere", s), npo.ops_gradient_sphere(s, Dvv, e["Dinv"], RR))
        assert close(O("divergence_sphere", v), npo.ops_divergence_sphere(v, Dvv, e["Dinv"], e["metdet"], RR))
        assert close(O("vorticity_sphere_vector", v), npo.ops_vorticity_sphere(v, Dvv, e["D"], e["metdet"], RR))
        assert close(O("divergence_sphere_wk", v), npo.ops_divergence_sphere_wk(v, Dvv, e["Dinv"], e["spheremp"], RR))
        assert close(O("laplace_simple", s), npo.ops_laplace_tensor(s, Dvv, e["Dinv"], e["spheremp"], None, RR))
        assert close(O("laplace_tensor", s), npo.ops_laplace_tensor(s, Dvv, e["Dinv"], e["spheremp"], e["tensorVisc"], RR))
        assert close(O("curl_sphere_wk_testcov", s), npo.ops_curl_sphere_wk_testcov(s, Dvv, e["D"], e["mp"], RR))
        assert close(O("grad_sphere_wk_testcov", s),
                     npo.ops_grad_sphere_wk_testcov(s, Dvv, e["D"], e["mp"], e["metinv"], e["metdet"], RR))
        assert close(O("vlaplace_sphere_wk_contra", v, nu_ratio=1.75),
                     npo.ops_vlaplace_sphere_wk_contra(v, Dvv, e["D"], e["Dinv"], e["mp"], e["spheremp"], e["metinv"],
                                                       e["metdet"], 1.75, RR))
        for rr_term in (1, 0):
            assert close(O("vlaplace_sphere_wk_cartesian", v, undamp_rr=rr_term),
                         npo.ops_vlaplace_sphere_wk_cartesian(v, Dvv, e["Dinv"], e["spheremp"], e["tensorVisc"],
                                                              e["vec_sph2cart"], RR, bool(rr_term)))


def test_polynomial_exactness_of_the_weak_laplacian(oracle):
    """On a flat, unit-metric element (D = Dinv = I, metdet = 1, spheremp = GLL weights w_a w_b) the weak
    Laplacian of a polynomial that vanishes with its normal derivative at the element edge is the mass-weighted
    strong Laplacian exactly (GLL quadrature is exact to degree 2N-1): checks sign, scaling and orientation
    of laplace_simple against calculus, not against another implementation."""
    for np_ in NPS:
        N = np_ - 1
        Dvv = cases.dvv_for(np_, "double" if np_ == 4 else "gll").T.copy()  # HOMME's orientation: Dvv[i][l] = l_i'(x_l)
        x = np.sort(np.real(np.roots(np.polyder(np.poly1d(np.polynomial.legendre.leg2poly([0] * N + [1])[::-1])))))
        x = np.concatenate(([-1.0], x, [1.0]))
        w = 2.0 / (N * (N + 1) * np.polynomial.legendre.Legendre.basis(N)(x) ** 2)
        e = {"Dinv": np.broadcast_to(np.eye(2), (np_, np_, 2, 2)).copy(), "spheremp": np.outer(w, w)}
        X, Y = np.meshgrid(x, x, indexing="ij")  # field[a][b]: a <-> x, b <-> y
        # u = (1-x^2)^2 (1-y^2)^2 has u = du/dn = 0 on the edge; degree 4 per direction <= N for np=8 only
        if np_ == 8:
            u = (1 - X ** 2) ** 2 * (1 - Y ** 2) ** 2
            uxx = (12 * X ** 2 - 4) * (1 - Y ** 2) ** 2
            uyy = (12 * Y ** 2 - 4) * (1 - X ** 2) ** 2
            lap = po.sphere_op(oracle, "laplace_simple", u, Dvv, e, 1.0)
            assert np.max(np.abs(lap - np.outer(w, w) * (uxx + uyy))) < 1e-12
        # any np: sum over the element of the weak Laplacian of any field is 0 (test function 1 has no gradient)
        s = cases.uniform((np_, np_), 91, -1, 1)
        assert abs(np.sum(po.sphere_op(oracle, "laplace_simple", s, Dvv, e, 1.0))) < 1e-12


# --------------------------------------------------------------------------------------- GPU: HIP vs the oracle
@pytest.mark.parametrize("np_", NPS)
def test_euler_step_oracle(oracle, np_):
    """oracle_euler_step (EulerStepFunctor.hpp:32-68, parity unpinned) is Qdp - dt * div(vstar * Qdp) with the
    divergence whose index mapping the first test pins; it reads only time level qn0 and the first qsize tracers; it is
    linear in Qdp; and a constant flux on the identity geometry leaves the tracer unchanged."""
    Dvv = cases.dvv_for(np_, "double" if np_ == 4 else "gll")
    g = geometry(np_, 2, 21)
    nlev, qsize_d, qsize, qn0, dt = 5, 3, 2, 1, 0.8
    vstar = cases.uniform((nlev, np_, np_, 2), 801, -3, 5)
    qdp = cases.uniform((qsize_d, 2, nlev, np_, np_), 802, 0.5, 2.0)
    e = elem(g, 1)
    got = po.euler_step(oracle, vstar, qdp, qsize, qn0, dt, Dvv, e["Dinv"], e["metdet"], RR)
    for q in range(qsize):
        for k in range(nlev):
            d = po.sphere_op(oracle, "divergence_sphere", vstar[k] * qdp[q, qn0, k][..., None], Dvv, e, RR)
            assert np.array_equal(got[q, k], qdp[q, qn0, k] * 1.0 + (-dt) * d)
    # the other time level and the tracers beyond qsize are not read
    qdp2 = qdp.copy()
    qdp2[:, 1 - qn0] = 7.0
    qdp2[qsize:] = -3.0
    assert np.array_equal(po.euler_step(oracle, vstar, qdp2, qsize, qn0, dt, Dvv, e["Dinv"], e["metdet"], RR), got)
    # linear in Qdp
    both = po.euler_step(oracle, vstar, 2.0 * qdp, qsize, qn0, dt, Dvv, e["Dinv"], e["metdet"], RR)
    assert close(both, 2.0 * got, 1e-15)
    # constant flux on the identity geometry: the divergence of a constant vanishes (Dvv^T rows sum to zero: see the
    # note on the reference's transposed Dvv in test_polynomial_exactness_of_the_weak_laplacian)
    ident = np.zeros((np_, np_, 2, 2))
    ident[..., 0, 0] = ident[..., 1, 1] = 1.0
    ones = np.ones((np_, np_))
    qc = np.full((1, 2, nlev, np_, np_), 1.5)
    vc = np.full((nlev, np_, np_, 2), 2.0)
    stay = po.euler_step(oracle, vc, qc, 1, 0, dt, np.ascontiguousarray(Dvv.T), ident, ones, RR)
    assert float(np.max(np.abs(stay - 1.5))) <= 1e-12


def _oracle_all(oracle, name, x, Dvv, g, e0, **kw):
    ne, nl = x.shape[:2]
    vout = po.SPHERE_OPS[name][1]
    out = np.zeros((ne, nl) + x.shape[2:4] + ((2,) if vout else ()))
    acc = kw.pop("out", None)
    for e in range(ne):
        for k in range(nl):
            out[e, k] = po.sphere_op(oracle, name, x[e, k], Dvv, elem(g, e0 + e), RR,
                                     out=None if acc is None else acc[e, k], **kw)
    return out


HIP_TO_ORACLE = {  # HIP operator name -> (oracle operator name, oracle kwargs)
    "divergence_sphere_wk": ("divergence_sphere_wk", {}), "laplace_simple": ("laplace_simple", {}),
    "laplace_tensor": ("laplace_tensor", {}), "laplace_tensor_replace": ("laplace_tensor_replace", {}),
    "curl_sphere_wk_testcov": ("curl_sphere_wk_testcov", {}),
    "grad_sphere_wk_testcov": ("grad_sphere_wk_testcov", {}),
    "vlaplace_sphere_wk_contra": ("vlaplace_sphere_wk_contra", dict(nu_ratio=1.75)),
    "vlaplace_sphere_wk_cartesian": ("vlaplace_sphere_wk_cartesian", dict(undamp_rr=1)),
    "vlaplace_sphere_wk_cartesian_damped": ("vlaplace_sphere_wk_cartesian", dict(undamp_rr=0)),
    "gradient_sphere": ("gradient_sphere", {}), "divergence_sphere": ("divergence_sphere", {}),
    "vorticity_sphere": ("vorticity_sphere_vector", {}),
}


@pytest.mark.gpu
@pytest.mark.parametrize("np_", NPS)
@pytest.mark.parametrize("name", list(HIP_TO_ORACLE))
def test_hip_operator_matches_oracle(oracle, np_, name):
    """caar_sphere_operator_ex against oracle/sphere_ops_oracle.c (parity unpinned, see the module docstring):
    5 elements of a 7-element geometry (e0 = 1), 11 levels (NP=4: a partly filled last tile), <= 1e-12."""
    import torch
    import tinman_sandbox_amd as tsa
    oname, okw = HIP_TO_ORACLE[name]
    Dvv = cases.dvv_for(np_, "double" if np_ == 4 else "gll")
    g = geometry(np_, 7, 12)
    ne, nl, e0 = 5, 11, 1
    vin = tsa.SPHERE_OPERATORS[name][1]
    x = cases.uniform((ne, nl, np_, np_) + ((2,) if vin else ()), 500 + np_, -3, 5)
    want = _oracle_all(oracle, oname, x, Dvv, g, e0, **okw)
    dev = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    got = tsa.sphere_operator_ex(name, torch.from_numpy(x).cuda(), dev, torch.from_numpy(Dvv).cuda(), RR,
                                 nu_ratio=okw.get("nu_ratio", 1.0), e0=e0)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    scale = float(np.max(np.abs(want)))
    assert float(np.max(np.abs(got - want))) <= 1e-12 * scale, (name, np_)


@pytest.mark.gpu
@pytest.mark.parametrize("np_", NPS)
def test_hip_update_operators_match_oracle(oracle, np_):
    import torch
    import tinman_sandbox_amd as tsa
    Dvv = cases.dvv_for(np_, "double" if np_ == 4 else "gll")
    g = geometry(np_, 3, 13)
    dev = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    dvv = torch.from_numpy(Dvv).cuda()
    ne, nl = 3, 9
    s = cases.uniform((ne, nl, np_, np_), 600, -3, 5)
    v = cases.uniform((ne, nl, np_, np_, 2), 601, -3, 5)
    acc_v = cases.uniform((ne, nl, np_, np_, 2), 602, -1, 1)
    acc_s = cases.uniform((ne, nl, np_, np_), 603, -1, 1)
    want = _oracle_all(oracle, "gradient_sphere_update", s, Dvv, g, 0, out=acc_v)
    got = tsa.sphere_operator_ex("gradient_sphere_update", torch.from_numpy(s).cuda(), dev, dvv, RR,
                                 out=torch.from_numpy(acc_v.copy()).cuda())
    assert float(np.max(np.abs(got.cpu().numpy() - want))) <= 1e-12 * float(np.max(np.abs(want)))
    want = _oracle_all(oracle, "divergence_sphere_update", v, Dvv, g, 0, out=acc_s, alpha=0.75, beta=-1.5)
    got = tsa.sphere_operator_ex("divergence_sphere_update", torch.from_numpy(v).cuda(), dev, dvv, RR,
                                 out=torch.from_numpy(acc_s.copy()).cuda(), alpha=0.75, beta=-1.5)
    assert float(np.max(np.abs(got.cpu().numpy() - want))) <= 1e-12 * float(np.max(np.abs(want)))


@pytest.mark.gpu
@pytest.mark.parametrize("np_,nlev,qsize_d,qsize", [(4, 72, 4, 4), (4, 11, 3, 2), (8, 9, 2, 2), (4, 6, 1, 1)])
def test_hip_euler_step_matches_oracle(oracle, np_, nlev, qsize_d, qsize):
    """caar_euler_step (EulerStepFunctor.hpp:32-68; parity unpinned) against oracle_euler_step: elements 1..4 of a
    6-element set, both Qdp time levels, fewer tracers than allocated, a partly filled last tile (NP=4, nlev 11 / 6)."""
    import torch
    import tinman_sandbox_amd as tsa
    Dvv = cases.dvv_for(np_, "double" if np_ == 4 else "gll")
    g = geometry(np_, 6, 22)
    dev = {k: torch.from_numpy(g[k]).cuda() for k in ("Dinv", "metdet", "rmetdet")}
    ne, e0, dt = 4, 1, 0.6
    vstar = cases.uniform((ne, nlev, np_, np_, 2), 811, -3, 5)
    qdp = cases.uniform((6, qsize_d, 2, nlev, np_, np_), 812, 0.5, 2.0)
    for qn0 in (0, 1):
        want = np.stack([po.euler_step(oracle, vstar[e], qdp[e0 + e], qsize, qn0, dt, Dvv, g["Dinv"][e0 + e],
                                       g["metdet"][e0 + e], RR) for e in range(ne)])
        got = tsa.euler_step(torch.from_numpy(vstar).cuda(), torch.from_numpy(qdp).cuda(), dev,
                             torch.from_numpy(Dvv).cuda(), qsize, qn0, dt, RR, e0=e0)
        torch.cuda.synchronize()
        err = float(np.max(np.abs(got.cpu().numpy() - want)))
        assert err <= 1e-12 * float(np.max(np.abs(want))), (np_, nlev, qn0, err)


@pytest.mark.gpu
def test_euler_step_validates_its_arguments():
    import ctypes as C
    import torch
    import tinman_sandbox_amd as tsa
    from tinman_sandbox_amd import caar as m
    lib = tsa.library().lib
    z = torch.zeros(4096, dtype=torch.float64, device="cuda")
    vp = C.c_void_p(z.data_ptr())
    geo = m._CaarOperatorGeometry()
    geo.Dinv = geo.metdet = geo.rmetdet = z.data_ptr()
    dims = m._CaarDims(4, 3, 2, 1, 2)
    call = lambda d, g_, e0, e1, qs, qn0: lib.caar_euler_step(C.byref(d), C.byref(g_), vp, e0, e1, qs, qn0, 1.0, 1.0, vp,
                                                              vp, vp, None)
    assert call(dims, geo, 0, 2, 2, 0) == 0
    assert call(dims, geo, 0, 3, 2, 0) == -1   # beyond num_elems
    assert call(dims, geo, 0, 2, 3, 0) == -1   # more tracers than qsize_d
    assert call(dims, geo, 0, 2, 2, 2) == -1   # Qdp has two time levels
    assert call(dims, m._CaarOperatorGeometry(), 0, 2, 2, 0) == -1  # no geometry
    assert call(m._CaarDims(5, 3, 2, 1, 2), geo, 0, 2, 2, 0) == -2   # unsupported np
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("np_", NPS)
def test_hip_single_level_forms_match_oracle(oracle, np_):
    """The reference's `_sl` forms (level_vectorized_ppscan/SphereOperators.hpp:17-227: gradient / divergence_wk /
    vorticity ... of ONE level) are the batched entry point with nlevels = 1 (NP=4: one live row of a tile, three dead
    ones).  Every operator code, one level, three elements; laplace_tensor_replace must also leave nothing but the
    result in its field."""
    import torch
    import tinman_sandbox_amd as tsa
    Dvv = cases.dvv_for(np_, "double" if np_ == 4 else "gll")
    g = geometry(np_, 3, 21)
    dev = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    for name, (oname, okw) in HIP_TO_ORACLE.items():
        vin = tsa.SPHERE_OPERATORS[name][1]
        x = cases.uniform((3, 1, np_, np_) + ((2,) if vin else ()), 900 + np_, -2, 3)
        want = _oracle_all(oracle, oname, x, Dvv, g, 0, **okw)
        xt = torch.from_numpy(x.copy()).cuda()
        got = tsa.sphere_operator_ex(name, xt, dev, torch.from_numpy(Dvv).cuda(), RR, nu_ratio=okw.get("nu_ratio", 1.0))
        torch.cuda.synchronize()
        if name == "laplace_tensor_replace":
            assert got.data_ptr() == xt.data_ptr()           # in place
        else:
            assert np.array_equal(xt.cpu().numpy(), x), name  # the input is not touched
        scale = float(np.max(np.abs(want)))
        assert float(np.max(np.abs(got.cpu().numpy() - want))) <= 1e-12 * scale, (name, np_)


@pytest.mark.gpu
def test_operator_ex_validates_its_arguments():
    import ctypes as C
    import torch
    import tinman_sandbox_amd as tsa
    from tinman_sandbox_amd import caar as m
    g = geometry(4, 2, 14)
    dev = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    x = torch.zeros((2, 3, 4, 4), dtype=torch.float64, device="cuda")
    dvv = torch.zeros((4, 4), dtype=torch.float64, device="cuda")
    with pytest.raises(tsa.caar.CaarError):  # laplace_tensor without tensorVisc
        tsa.sphere_operator_ex("laplace_tensor", x, {k: dev[k] for k in ("Dinv", "spheremp")}, dvv, RR)
    with pytest.raises(tsa.caar.CaarError):  # element range beyond the geometry
        tsa.sphere_operator_ex("laplace_simple", x, dev, dvv, RR, e0=1)
    lib = tsa.library().lib
    dims = m._CaarDims(4, 3, 1, 1, 2)
    geo = m._CaarOperatorGeometry()
    sc = m._CaarOperatorScalars(1.0, 1.0, 0.0, 1.0)
    assert lib.caar_sphere_operator_ex(C.byref(dims), C.byref(geo), C.c_void_p(dvv.data_ptr()), 99, 0, 2, 3,
                                       C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr()), C.byref(sc), None) == -1
    # only the in-place code may come without an input pointer
    for n in ("Dinv", "spheremp", "tensorVisc"):
        setattr(geo, n, dev[n].data_ptr())
    code_lt, code_ltr = tsa.SPHERE_OPERATORS["laplace_tensor"][0], tsa.SPHERE_OPERATORS["laplace_tensor_replace"][0]
    assert lib.caar_sphere_operator_ex(C.byref(dims), C.byref(geo), C.c_void_p(dvv.data_ptr()), code_lt, 0, 2, 3, None,
                                       C.c_void_p(x.data_ptr()), C.byref(sc), None) == -1
    assert lib.caar_sphere_operator_ex(C.byref(dims), C.byref(geo), C.c_void_p(dvv.data_ptr()), code_ltr, 0, 2, 3, None,
                                       C.c_void_p(x.data_ptr()), C.byref(sc), None) == 0
    torch.cuda.synchronize()


@pytest.mark.gpu
@pytest.mark.parametrize("np_,nlev", [(4, 72), (8, 72), (4, 26)])
def test_standalone_vertical_integrals_are_bit_identical_to_the_oracle(oracle, np_, nlev):
    """caar_preq_hydrostatic / caar_preq_omega_ps (reference compute_and_apply_rhs.hpp:11-17, P:280-352): one thread per
    column in the reference's order, no contraction, IEEE division => the oracle's (== the reference's) bits."""
    import ctypes as C
    import torch
    import tinman_sandbox_amd as tsa
    from tinman_sandbox_amd import caar as m
    ne = 3
    phis = cases.uniform((ne, np_, np_), 700, 0, 3e4)
    Tv = cases.uniform((ne, nlev, np_, np_), 701, 200, 310)
    p = np.cumsum(cases.uniform((ne, nlev, np_, np_), 702, 500, 1500), axis=1)
    dp = cases.uniform((ne, nlev, np_, np_), 703, 500, 1500)
    vg = cases.uniform((ne, nlev, np_, np_), 704, -50, 50)
    dd = cases.uniform((ne, nlev, np_, np_), 705, -5, 5)
    phi_w, om_w = np.zeros_like(Tv), np.zeros_like(Tv)
    P = po._ptr
    oracle.lib.oracle_preq_hydrostatic.argtypes = [C.c_int, C.c_int] + [po._dp] * 4 + [C.c_double, po._dp]
    oracle.lib.oracle_preq_omega_ps.argtypes = [C.c_int, C.c_int] + [po._dp] * 4
    for e in range(ne):
        out = np.zeros((nlev, np_, np_))
        oracle.lib.oracle_preq_hydrostatic(np_, nlev, P(phis[e]), P(Tv[e]), P(p[e].copy()), P(dp[e]), 287.04, P(out))
        phi_w[e] = out
        out = np.zeros((nlev, np_, np_))
        oracle.lib.oracle_preq_omega_ps(np_, nlev, P(p[e].copy()), P(vg[e]), P(dd[e]), P(out))
        om_w[e] = out
    lib = tsa.library().lib
    dims = m._CaarDims(np_, nlev, 1, 1, ne)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in
         dict(phis=phis, Tv=Tv, p=p, dp=dp, vg=vg, dd=dd).items()}
    phi = torch.empty_like(t["Tv"])
    om = torch.empty_like(t["Tv"])
    V = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    assert lib.caar_preq_hydrostatic(C.byref(dims), ne, V(t["phis"]), V(t["Tv"]), V(t["p"]), V(t["dp"]), 287.04, V(phi), None) == 0
    assert lib.caar_preq_omega_ps(C.byref(dims), ne, V(t["p"]), V(t["vg"]), V(t["dd"]), V(om), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(phi.cpu().numpy(), phi_w)
    assert np.array_equal(om.cpu().numpy(), om_w)
