"""ctypes bindings for the CPU checkers under oracle/.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package (tinman_sandbox_amd/) never does.

  Oracle      -> oracle/libcaar_oracle.so   (this repo's C restatement, caar_oracle.c)
  Reference   -> oracle/_ref/libref_caar_np<NP>_nlev<NLEV>.so (the reference's own
                 C++ path compiled from /root/reference by oracle/Makefile)

Arrays are passed as a dict name -> contiguous float64 numpy array, names and
order being the members of Homme::Arrays
(compute_and_apply_rhs_test/cxx/pointers_only/data_structures.hpp:18-44).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

ARRAY_NAMES = (
    "elem_D", "elem_Dinv", "elem_fcor", "elem_spheremp", "elem_metdet", "elem_rmetdet",
    "elem_state_dp3d", "elem_state_v", "elem_state_T", "elem_state_phis", "elem_state_Qdp",
    "elem_derived_eta_dot_dpdn", "elem_derived_omega_p", "elem_derived_phi",
    "elem_derived_pecnd", "elem_derived_vn0",
)

_dp = C.POINTER(C.c_double)


def array_shapes(np_, nlev, qsize_d, timelevels, ne):
    """Element-major C layouts, data_structures.cpp:14-31."""
    return {
        "elem_D": (ne, np_, np_, 2, 2),
        "elem_Dinv": (ne, np_, np_, 2, 2),
        "elem_fcor": (ne, np_, np_),
        "elem_spheremp": (ne, np_, np_),
        "elem_metdet": (ne, np_, np_),
        "elem_rmetdet": (ne, np_, np_),
        "elem_state_dp3d": (ne, timelevels, nlev, np_, np_),
        "elem_state_v": (ne, timelevels, nlev, np_, np_, 2),
        "elem_state_T": (ne, timelevels, nlev, np_, np_),
        "elem_state_phis": (ne, np_, np_),
        "elem_state_Qdp": (ne, qsize_d, 2, nlev, np_, np_),
        "elem_derived_eta_dot_dpdn": (ne, nlev + 1, np_, np_),
        "elem_derived_omega_p": (ne, nlev, np_, np_),
        "elem_derived_phi": (ne, nlev, np_, np_),
        "elem_derived_pecnd": (ne, nlev, np_, np_),
        "elem_derived_vn0": (ne, nlev, np_, np_, 2),
    }


def alloc_arrays(np_, nlev, qsize_d, timelevels, ne):
    return {k: np.zeros(s, dtype=np.float64)
            for k, s in array_shapes(np_, nlev, qsize_d, timelevels, ne).items()}


class _OracleArrays(C.Structure):
    _fields_ = [(n, _dp) for n in ARRAY_NAMES]


class _OracleParams(C.Structure):
    _fields_ = [
        ("np", C.c_int), ("nlev", C.c_int), ("qsize_d", C.c_int), ("timelevels", C.c_int),
        ("nets", C.c_int), ("nete", C.c_int),
        ("n0", C.c_int), ("np1", C.c_int), ("nm1", C.c_int), ("qn0", C.c_int),
        ("dt2", C.c_double),
        ("rrearth", C.c_double), ("eta_ave_w", C.c_double), ("Rwater_vapor", C.c_double),
        ("Rgas", C.c_double), ("kappa", C.c_double),
        ("ps0", C.c_double), ("hyai0", C.c_double),
        ("Dvv", _dp),
        ("rsplit", C.c_int), ("hybi", _dp),
    ]


def _ptr(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_dp)


def default_scalars(nlev):
    """Constants/Control/HVCoord::init_data, data_structures.cpp:117-150."""
    Rgas, cp = 287.04, 1005.0
    return dict(nets=0, nete=None, n0=0, np1=1, nm1=2, qn0=0, dt2=1.0,
                rrearth=1.0 / 6.376e6, eta_ave_w=1.0, Rwater_vapor=461.5, Rgas=Rgas,
                kappa=Rgas / cp, ps0=10.0,
                hyai=np.array([nlev + 1 - i for i in range(nlev + 1)], dtype=np.float64))


def build(ref=True):
    """Compile the checkers (gcc / g++ / flang) via oracle/Makefile."""
    subprocess.run(["make", "-C", HERE, "liboracle"], check=True, capture_output=True)
    if ref and os.path.isdir("/root/reference/compute_and_apply_rhs_test"):
        subprocess.run(["make", "-C", HERE, "ref"], check=True, capture_output=True)


class Oracle:
    """This repo's C restatement (oracle/caar_oracle.c)."""

    def __init__(self, path=None):
        path = path or os.path.join(HERE, "libcaar_oracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = self.lib = C.CDLL(path)
        L.oracle_compute_and_apply_rhs.argtypes = [C.POINTER(_OracleArrays), C.POINTER(_OracleParams)]
        L.oracle_compute_and_apply_rhs.restype = C.c_int
        L.oracle_compute_and_apply_rhs_repeat.argtypes = [C.POINTER(_OracleArrays), C.POINTER(_OracleParams), C.c_int]
        L.oracle_compute_and_apply_rhs_repeat.restype = C.c_int
        L.oracle_state_norms.argtypes = [C.POINTER(_OracleArrays), C.POINTER(_OracleParams), _dp]
        L.oracle_compute_norm.argtypes = [_dp, C.c_long]
        L.oracle_compute_norm.restype = C.c_double
        L.oracle_init_arrays.argtypes = [C.POINTER(_OracleArrays)] + [C.c_int] * 5
        L.oracle_init_dvv_np4.argtypes = [_dp, C.c_int]
        L.oracle_init_dvv_gll.argtypes = [C.c_int, _dp]
        L.oracle_gradient_sphere.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double, _dp]
        L.oracle_divergence_sphere.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_double, _dp]
        L.oracle_vorticity_sphere.argtypes = [C.c_int, _dp, _dp, _dp, _dp, C.c_double, _dp]
        d, i, f = _dp, C.c_int, C.c_double
        for name, args in {
                "oracle_k_gradient_sphere": [i, d, d, d, f, d], "oracle_gradient_sphere_update": [i, d, d, d, f, d],
                "oracle_k_divergence_sphere": [i, d, d, d, d, f, d],
                "oracle_divergence_sphere_update": [i, f, f, d, d, d, d, f, d],
                "oracle_k_vorticity_sphere_vector": [i, d, d, d, d, f, d],
                "oracle_divergence_sphere_wk": [i, d, d, d, d, f, d], "oracle_laplace_simple": [i, d, d, d, d, f, d],
                "oracle_laplace_tensor": [i, d, d, d, d, d, f, d], "oracle_laplace_tensor_replace": [i, d, d, d, d, f, d], "oracle_curl_sphere_wk_testcov": [i, d, d, d, d, f, d],
                "oracle_grad_sphere_wk_testcov": [i, d, d, d, d, d, d, f, d],
                "oracle_vlaplace_sphere_wk_cartesian": [i, d, d, d, d, d, d, f, i, d],
                "oracle_vlaplace_sphere_wk_contra": [i, d, d, d, d, d, d, d, d, f, f, d],
                "oracle_euler_step": [i, i, i, i, f, d, d, d, d, d, f, d]}.items():
            getattr(L, name).argtypes = args
            getattr(L, name).restype = None

    @staticmethod
    def _arrays(arrs):
        return _OracleArrays(*[_ptr(arrs[n]) for n in ARRAY_NAMES])

    def _params(self, arrs, Dvv, sc):
        ne, tl, nlev, np_, _ = arrs["elem_state_dp3d"].shape
        qd = arrs["elem_state_Qdp"].shape[1]
        nete = ne if sc.get("nete") is None else sc["nete"]
        self._keep = np.ascontiguousarray(Dvv, dtype=np.float64)
        rsplit = int(sc.get("rsplit", 1))
        self._hybi = None
        if rsplit == 0:
            self._hybi = np.ascontiguousarray(sc["hybi"], dtype=np.float64)
            assert self._hybi.size == nlev + 1
        return _OracleParams(np_, nlev, qd, tl, sc["nets"], nete, sc["n0"], sc["np1"], sc["nm1"],
                             sc["qn0"], sc["dt2"], sc["rrearth"], sc["eta_ave_w"],
                             sc["Rwater_vapor"], sc["Rgas"], sc["kappa"], sc["ps0"],
                             float(sc["hyai"][0]), _ptr(self._keep), rsplit,
                             _ptr(self._hybi) if rsplit == 0 else None)

    def compute_and_apply_rhs(self, arrs, Dvv, sc, reps=1):
        """reps > 1: that many back-to-back calls inside ONE C call (no Python, no GIL between calls)."""
        a, p = self._arrays(arrs), self._params(arrs, Dvv, sc)
        rc = self.lib.oracle_compute_and_apply_rhs_repeat(C.byref(a), C.byref(p), int(reps))
        if rc != 0:
            raise RuntimeError("oracle_compute_and_apply_rhs rc=%d" % rc)

    def state_norms(self, arrs, Dvv, sc):
        out = np.zeros(3)
        a, p = self._arrays(arrs), self._params(arrs, Dvv, sc)
        self.lib.oracle_state_norms(C.byref(a), C.byref(p), _ptr(out))
        return out

    def compute_norm(self, field):
        f = np.ascontiguousarray(field, dtype=np.float64).ravel()
        return self.lib.oracle_compute_norm(_ptr(f), f.size)

    def init_arrays(self, np_, nlev, qsize_d, timelevels, ne):
        arrs = alloc_arrays(np_, nlev, qsize_d, timelevels, ne)
        a = self._arrays(arrs)
        self.lib.oracle_init_arrays(C.byref(a), np_, nlev, qsize_d, timelevels, ne)
        return arrs

    def dvv_np4(self, f32_rounded=False):
        d = np.zeros((4, 4))
        self.lib.oracle_init_dvv_np4(_ptr(d), int(f32_rounded))
        return d

    def dvv_gll(self, np_):
        d = np.zeros((np_, np_))
        self.lib.oracle_init_dvv_gll(np_, _ptr(d))
        return d

    def gradient_sphere(self, s, Dvv, Dinv, rrearth):
        np_ = s.shape[0]
        out = np.zeros((np_, np_, 2))
        self.lib.oracle_gradient_sphere(np_, _ptr(np.ascontiguousarray(s)), _ptr(np.ascontiguousarray(Dvv)),
                                        _ptr(np.ascontiguousarray(Dinv)), rrearth, _ptr(out))
        return out

    def divergence_sphere(self, v, Dvv, Dinv, metdet, rmetdet, rrearth):
        np_ = v.shape[0]
        out = np.zeros((np_, np_))
        self.lib.oracle_divergence_sphere(np_, _ptr(np.ascontiguousarray(v)), _ptr(np.ascontiguousarray(Dvv)),
                                          _ptr(np.ascontiguousarray(Dinv)), _ptr(np.ascontiguousarray(metdet)),
                                          _ptr(np.ascontiguousarray(rmetdet)), rrearth, _ptr(out))
        return out

    def vorticity_sphere(self, v, Dvv, D, rmetdet, rrearth):
        np_ = v.shape[0]
        out = np.zeros((np_, np_))
        self.lib.oracle_vorticity_sphere(np_, _ptr(np.ascontiguousarray(v)), _ptr(np.ascontiguousarray(Dvv)),
                                         _ptr(np.ascontiguousarray(D)), _ptr(np.ascontiguousarray(rmetdet)),
                                         rrearth, _ptr(out))
        return out


# ---- sphere operators next to the CAAR path (sphere_ops_oracle.c; PARITY UNPINNED, see its header) ----------
# name -> (input is a vector field, output is a vector field, geometry arrays in call order)
SPHERE_OPS = {
    "gradient_sphere": (False, True, ("Dinv",)),
    "divergence_sphere": (True, False, ("Dinv", "metdet")),
    "vorticity_sphere_vector": (True, False, ("D", "metdet")),
    "divergence_sphere_wk": (True, False, ("Dinv", "spheremp")),
    "laplace_simple": (False, False, ("Dinv", "spheremp")),
    "laplace_tensor": (False, False, ("Dinv", "spheremp", "tensorVisc")),
    "laplace_tensor_replace": (False, False, ("Dinv", "spheremp", "tensorVisc")),
    "curl_sphere_wk_testcov": (False, True, ("D", "mp")),
    "grad_sphere_wk_testcov": (False, True, ("D", "mp", "metinv", "metdet")),
    "vlaplace_sphere_wk_cartesian": (True, True, ("Dinv", "spheremp", "tensorVisc", "vec_sph2cart")),
    "vlaplace_sphere_wk_contra": (True, True, ("D", "Dinv", "mp", "spheremp", "metinv", "metdet")),
    "gradient_sphere_update": (False, True, ("Dinv",)),
    "divergence_sphere_update": (True, False, ("Dinv", "metdet")),
}


def sphere_op(O, name, x, Dvv, geo, rrearth, out=None, alpha=1.0, beta=0.0, nu_ratio=1.0, undamp_rr=1):
    """One level of one element through the C oracle.  x: [np][np] or [np][np][2]; geo: dict of that element's
    geometry arrays ([np][np], [np][np][2][2], vec_sph2cart [np][np][3][2]); `out` is the array the *_update
    operators accumulate into (copied, not modified)."""
    vin, vout, gnames = SPHERE_OPS[name]
    np_ = x.shape[0]
    x = np.ascontiguousarray(x, dtype=np.float64)
    Dvv = np.ascontiguousarray(Dvv, dtype=np.float64)
    g = [np.ascontiguousarray(geo[n], dtype=np.float64) for n in gnames]
    res = np.zeros((np_, np_, 2) if vout else (np_, np_)) if out is None else np.array(out, dtype=np.float64, order="C")
    L = O.lib
    P = _ptr
    if name == "gradient_sphere":
        L.oracle_k_gradient_sphere(np_, P(x), P(Dvv), P(g[0]), rrearth, P(res))
    elif name == "gradient_sphere_update":
        L.oracle_gradient_sphere_update(np_, P(x), P(Dvv), P(g[0]), rrearth, P(res))
    elif name == "divergence_sphere":
        L.oracle_k_divergence_sphere(np_, P(x), P(Dvv), P(g[0]), P(g[1]), rrearth, P(res))
    elif name == "divergence_sphere_update":
        L.oracle_divergence_sphere_update(np_, alpha, beta, P(x), P(Dvv), P(g[0]), P(g[1]), rrearth, P(res))
    elif name == "vorticity_sphere_vector":
        L.oracle_k_vorticity_sphere_vector(np_, P(x), P(Dvv), P(g[0]), P(g[1]), rrearth, P(res))
    elif name == "divergence_sphere_wk":
        L.oracle_divergence_sphere_wk(np_, P(x), P(Dvv), P(g[0]), P(g[1]), rrearth, P(res))
    elif name == "laplace_simple":
        L.oracle_laplace_simple(np_, P(x), P(Dvv), P(g[0]), P(g[1]), rrearth, P(res))
    elif name == "laplace_tensor":
        L.oracle_laplace_tensor(np_, P(x), P(Dvv), P(g[0]), P(g[1]), P(g[2]), rrearth, P(res))
    elif name == "laplace_tensor_replace":  # in place: the result replaces (a copy of) the input
        res = np.array(x, dtype=np.float64, order="C")
        L.oracle_laplace_tensor_replace(np_, P(Dvv), P(g[0]), P(g[1]), P(g[2]), rrearth, P(res))
    elif name == "curl_sphere_wk_testcov":
        L.oracle_curl_sphere_wk_testcov(np_, P(x), P(Dvv), P(g[0]), P(g[1]), rrearth, P(res))
    elif name == "grad_sphere_wk_testcov":
        L.oracle_grad_sphere_wk_testcov(np_, P(x), P(Dvv), P(g[0]), P(g[1]), P(g[2]), P(g[3]), rrearth, P(res))
    elif name == "vlaplace_sphere_wk_cartesian":
        L.oracle_vlaplace_sphere_wk_cartesian(np_, P(x), P(Dvv), P(g[0]), P(g[1]), P(g[2]), P(g[3]), rrearth,
                                              int(undamp_rr), P(res))
    elif name == "vlaplace_sphere_wk_contra":
        L.oracle_vlaplace_sphere_wk_contra(np_, P(x), P(Dvv), P(g[0]), P(g[1]), P(g[2]), P(g[3]), P(g[4]), P(g[5]),
                                           nu_ratio, rrearth, P(res))
    else:
        raise KeyError(name)
    return res


def euler_step(O, vstar, qdp, qsize, qn0, dt, Dvv, Dinv, metdet, rrearth):
    """EulerStepFunctor.hpp:32-68 of ONE element through the C oracle (PARITY UNPINNED): vstar [nlev][np][np][2],
    qdp [qsize_d][2][nlev][np][np] -> qtens [qsize][nlev][np][np]."""
    nlev, np_ = vstar.shape[0], vstar.shape[1]
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (vstar, qdp, Dvv, Dinv, metdet)]
    qtens = np.zeros((qsize, nlev, np_, np_))
    O.lib.oracle_euler_step(np_, nlev, qsize, qn0, dt, _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), _ptr(a[3]), _ptr(a[4]), rrearth,
                            _ptr(qtens))
    return qtens


def ref_lib_path(np_, nlev):
    return os.path.join(HERE, "_ref", "libref_caar_np%d_nlev%d.so" % (np_, nlev))


def have_reference(np_=4, nlev=72):
    return os.path.exists(ref_lib_path(np_, nlev))


class Reference:
    """The reference's own C++ path (oracle/_ref/, built by oracle/Makefile from
    /root/reference; qsize_d=1, timelevels=3 are fixed by its config.h.in)."""

    def __init__(self, np_=4, nlev=72):
        self.np, self.nlev = np_, nlev
        L = self.lib = C.CDLL(ref_lib_path(np_, nlev))
        dims = (C.c_int * 4)()
        L.ref_dims(dims)
        assert (dims[0], dims[1]) == (np_, nlev)
        self.qsize_d, self.timelevels = dims[2], dims[3]
        pp = _dp * 16
        L.ref_compute_and_apply_rhs.argtypes = ([pp] + [C.c_int] * 6 + [C.c_double] * 7 + [_dp, _dp])
        self.has_repeat = hasattr(L, "ref_compute_and_apply_rhs_repeat")  # (libraries built before round 5 lack it)
        if self.has_repeat:
            L.ref_compute_and_apply_rhs_repeat.argtypes = L.ref_compute_and_apply_rhs.argtypes + [C.c_int]
        L.ref_init_data.argtypes = [C.c_int, pp, _dp, _dp, _dp]
        L.ref_sphere_operator.argtypes = [C.c_int, _dp, _dp, pp, C.c_int, C.c_double, _dp]
        L.ref_compute_norm.argtypes = [_dp, C.c_int]
        L.ref_compute_norm.restype = C.c_double

    @staticmethod
    def _pp(arrs):
        return (_dp * 16)(*[_ptr(arrs[n]) for n in ARRAY_NAMES])

    def compute_and_apply_rhs(self, arrs, Dvv, sc, reps=1):
        """reps > 1: the reference driver's loop (main.cpp:113-121) inside ONE C call (no Python, no GIL between calls)."""
        ne = arrs["elem_state_dp3d"].shape[0]
        nete = ne if sc.get("nete") is None else sc["nete"]
        Dvv = np.ascontiguousarray(Dvv, dtype=np.float64)
        hyai = np.ascontiguousarray(sc["hyai"], dtype=np.float64)
        assert hyai.size == self.nlev + 1 and Dvv.size == self.np ** 2
        args = (self._pp(arrs), sc["nets"], nete, sc["n0"], sc["np1"], sc["nm1"], sc["qn0"], sc["dt2"], sc["rrearth"],
                sc["eta_ave_w"], sc["Rwater_vapor"], sc["Rgas"], sc["kappa"], sc["ps0"], _ptr(hyai), _ptr(Dvv))
        if reps != 1 and self.has_repeat:
            self.lib.ref_compute_and_apply_rhs_repeat(*(args + (int(reps),)))
            return
        for _ in range(reps):
            self.lib.ref_compute_and_apply_rhs(*args)

    def init_data(self, ne):
        arrs = alloc_arrays(self.np, self.nlev, self.qsize_d, self.timelevels, ne)
        scal = np.zeros(12)
        hyai = np.zeros(self.nlev + 1)
        Dvv = np.zeros((self.np, self.np))
        self.lib.ref_init_data(ne, self._pp(arrs), _ptr(scal), _ptr(hyai), _ptr(Dvv))
        sc = dict(nets=0, nete=None, n0=int(scal[8]), np1=int(scal[9]), nm1=int(scal[10]),
                  qn0=int(scal[11]), dt2=scal[6], rrearth=scal[0], eta_ave_w=scal[1],
                  Rwater_vapor=scal[3], Rgas=scal[4], kappa=scal[5], ps0=scal[7], hyai=hyai)
        return arrs, Dvv, sc

    def sphere_operator(self, which, x, arrs, ie, rrearth, Dvv):
        np_ = self.np
        out = np.zeros((np_, np_, 2) if which == 0 else (np_, np_))
        x = np.ascontiguousarray(x, dtype=np.float64)
        Dvv = np.ascontiguousarray(Dvv, dtype=np.float64)
        self.lib.ref_sphere_operator(which, _ptr(x), _ptr(out), self._pp(arrs), ie, rrearth, _ptr(Dvv))
        return out

    def compute_norm(self, field):
        f = np.ascontiguousarray(field, dtype=np.float64).ravel()
        return self.lib.ref_compute_norm(_ptr(f), f.size)


def run_fortran_driver(arrs, Dvv, sc, native=False, exe_name="fortran_driver"):
    """Run oracle/_ref/fortran_driver (reference Fortran routine_mod, NP=4 NLEV=72,
    physical constants fixed by physical_constants.F90) on `arrs`; returns the
    mutated arrays.  `arrs` itself is left untouched.  native=True returns instead a
    dict with "c_<name>" (C++ layout, via the driver's explicit index loops) and
    "f_<name>" (the same array in Fortran's native element order, shaped as a C-ordered
    array with the Fortran first index last) for the layout-kernel fixtures.
    exe_name="fortran_driver_hip": the same driver program linked against the HIP drop-in module
    (host/fortran/routine_mod_hip.F90) instead of the reference's routine_mod — the thing under test then."""
    import tempfile
    exe = os.path.join(HERE, "_ref", exe_name)
    ne, tl, nlev, np_, _ = arrs["elem_state_dp3d"].shape
    with tempfile.TemporaryDirectory() as td:
        fin, fout, fnat = os.path.join(td, "in.bin"), os.path.join(td, "out.bin"), os.path.join(td, "nat.bin")
        with open(fin, "wb") as f:
            np.array([ne, sc["n0"], sc["np1"], sc["nm1"], sc["qn0"]], dtype=np.int32).tofile(f)
            np.array([sc["dt2"], sc["eta_ave_w"], sc["ps0"]], dtype=np.float64).tofile(f)
            np.ascontiguousarray(sc["hyai"], dtype=np.float64).tofile(f)
            np.ascontiguousarray(Dvv, dtype=np.float64).tofile(f)
            for n in ARRAY_NAMES:
                arrs[n].tofile(f)
        subprocess.run([exe, fin, fout] + ([fnat] if native else []), check=True)
        out = {}
        names = ("elem_state_dp3d", "elem_state_v", "elem_state_T", "elem_derived_eta_dot_dpdn",
                 "elem_derived_omega_p", "elem_derived_phi", "elem_derived_vn0")
        with open(fout, "rb") as f:
            for n in names:
                out[n] = np.fromfile(f, dtype=np.float64, count=arrs[n].size).reshape(arrs[n].shape)
        if not native:
            return out
        res = {"c_" + n: out[n] for n in names}
        fshape = {
            "elem_state_dp3d": (ne, tl, nlev, np_, np_), "elem_state_v": (ne, tl, nlev, 2, np_, np_),
            "elem_state_T": (ne, tl, nlev, np_, np_), "elem_derived_eta_dot_dpdn": (ne, nlev + 1, np_, np_),
            "elem_derived_omega_p": (ne, nlev, np_, np_), "elem_derived_phi": (ne, nlev, np_, np_),
            "elem_derived_vn0": (ne, nlev, 2, np_, np_), "elem_state_Qdp": (ne, 2, 1, nlev, np_, np_),
            "elem_D": (ne, 2, 2, np_, np_), "elem_fcor": (ne, np_, np_)}
        with open(fnat, "rb") as f:
            for n in names + ("elem_state_Qdp", "elem_D", "elem_fcor"):
                cnt = int(np.prod(fshape[n]))
                res["f_" + n] = np.fromfile(f, dtype=np.float64, count=cnt).reshape(fshape[n])
        for n in ("elem_state_Qdp", "elem_D", "elem_fcor"):
            res["c_" + n] = arrs[n].copy()
    return res
