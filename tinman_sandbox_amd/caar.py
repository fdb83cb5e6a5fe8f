"""Host-side mirror of the reference's interface for compute_and_apply_rhs, on top of
the C ABI (include/caar.h, csrc/libcaar_hip.so).

Names follow the reference's C++ driver so that tests read like its own code:
  TestData{arrays, constants, control, deriv, hvcoord}, init_data(),
  update_time_levels(), compute_and_apply_rhs(data), print_results_2norm(data)
  (compute_and_apply_rhs_test/cxx/pointers_only/data_structures.hpp:10-89,
   compute_and_apply_rhs.hpp:9-22, main.cpp:96-131).

torch is used for what it is good at here — owning device memory and streams —
and nothing else: the arithmetic runs in the hand-written HIP kernels.  There is
no CPU fallback: without the built library or without a GPU the calls raise.
"""
import ctypes as C
import math
import os

import numpy as np
import torch  # imported BEFORE the HIP library is loaded so both share one HIP runtime

_HERE = os.path.dirname(os.path.abspath(__file__))
# CAAR_LIBRARY=debug loads the -DCAAR_DEBUG build (dp3d(np1) > 0 checked by the kernels, caar_debug_dp3d_violations);
# CAAR_LIBRARY=extra the -DCAAR_EXTRA_NLEV=1 build (shapes specialised for seven more level counts, the Eulerian form
# beyond 128 levels: include/caar.h caar_supported_ex)
LIB_PATH = os.path.join(_HERE, "csrc", {"debug": "libcaar_hip_debug.so", "extra": "libcaar_hip_extra.so"}.get(
    os.environ.get("CAAR_LIBRARY", ""), "libcaar_hip.so"))
if os.environ.get("CAAR_LIBRARY_PATH"):  # an explicitly named build (A/B of compile-time choices)
    LIB_PATH = os.environ["CAAR_LIBRARY_PATH"]

# member order of Homme::Arrays (data_structures.hpp:18-44) == CaarArrays
ARRAY_NAMES = (
    "elem_D", "elem_Dinv", "elem_fcor", "elem_spheremp", "elem_metdet", "elem_rmetdet",
    "elem_state_dp3d", "elem_state_v", "elem_state_T", "elem_state_phis", "elem_state_Qdp",
    "elem_derived_eta_dot_dpdn", "elem_derived_omega_p", "elem_derived_phi",
    "elem_derived_pecnd", "elem_derived_vn0",
)
MUTATED = ("elem_state_dp3d", "elem_state_v", "elem_state_T", "elem_derived_eta_dot_dpdn",
           "elem_derived_omega_p", "elem_derived_phi", "elem_derived_vn0")

_dp = C.POINTER(C.c_double)


class _CaarArrays(C.Structure):
    _fields_ = [(n, _dp) for n in ARRAY_NAMES]


class _CaarDims(C.Structure):
    _fields_ = [("np", C.c_int), ("nlev", C.c_int), ("qsize_d", C.c_int),
                ("timelevels", C.c_int), ("num_elems", C.c_int)]


class _CaarParams(C.Structure):
    _fields_ = [("nets", C.c_int), ("nete", C.c_int), ("n0", C.c_int), ("np1", C.c_int),
                ("nm1", C.c_int), ("qn0", C.c_int), ("dt2", C.c_double), ("rrearth", C.c_double),
                ("eta_ave_w", C.c_double), ("Rwater_vapor", C.c_double), ("Rgas", C.c_double),
                ("kappa", C.c_double), ("ps0", C.c_double), ("hyai0", C.c_double), ("Dvv", _dp),
                ("rsplit", C.c_int), ("hybi", _dp), ("hybi_dev", _dp)]


class CaarError(RuntimeError):
    pass


class _CaarPlacement(C.Structure):
    _fields_ = [("policy", C.c_int), ("pool_bytes", C.c_longlong), ("max_free_fraction", C.c_double)]


PLACE_DEFAULT, PLACE_SPREAD, PLACE_MALLOC = 0, 1, 2


def placement(policy="spread", pool_gib=0, max_free_fraction=0.0):
    """CaarPlacement (include/caar.h) for caar_arrays_alloc_ex / caar_create_ex: policy "default" | "spread" | "malloc",
    the bound of the temporary pool in GiB (0: the library's default) and the largest share of the free device memory
    the pool may take (0: one half)."""
    code = {"default": PLACE_DEFAULT, "spread": PLACE_SPREAD, "malloc": PLACE_MALLOC}[policy]
    return _CaarPlacement(code, int(pool_gib * (1 << 30)), float(max_free_fraction))


class _CaarOperatorGeometry(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("D", "Dinv", "metdet", "rmetdet", "spheremp", "mp", "metinv", "tensorVisc",
                                           "vec_sph2cart")]


class _CaarOperatorScalars(C.Structure):
    _fields_ = [("rrearth", C.c_double), ("alpha", C.c_double), ("beta", C.c_double), ("nu_ratio", C.c_double)]


# caar_sphere_operator_ex `which` codes (include/caar.h): name -> (code, vector input, vector output)
SPHERE_OPERATORS = {
    "gradient_sphere": (0, False, True), "divergence_sphere": (1, True, False), "vorticity_sphere": (2, True, False),
    "divergence_sphere_wk": (3, True, False), "laplace_simple": (4, False, False), "laplace_tensor": (5, False, False),
    "curl_sphere_wk_testcov": (6, False, True), "grad_sphere_wk_testcov": (7, False, True),
    "vlaplace_sphere_wk_contra": (8, True, True), "vlaplace_sphere_wk_cartesian": (9, True, True),
    "gradient_sphere_update": (10, False, True), "divergence_sphere_update": (11, True, False),
    "vlaplace_sphere_wk_cartesian_damped": (12, True, True),
    "laplace_tensor_replace": (13, False, False),   # in place: `field` is overwritten and returned
}


class CaarLibrary:
    """ctypes view of libcaar_hip.so; every symbol of include/caar.h and include/caar_tuning.h is bound here."""

    # the frozen boundary, include/caar.h (ABI 6) — tests/test_host.py holds this list against the header and INTEGRATION.md
    BOUNDARY_SYMBOLS = (
        "caar_supported", "caar_supported_ex", "caar_abi_version", "caar_device_count", "caar_strerror", "caar_array_len",
        "caar_algorithmic_bytes", "caar_launch", "caar_launch_steps", "caar_sphere_operator", "caar_sphere_operator_range",
        "caar_sphere_operator_ex", "caar_euler_step", "caar_preq_hydrostatic", "caar_preq_omega_ps", "caar_sphere_operator_host",
        "caar_preq_hydrostatic_host", "caar_preq_omega_ps_host", "caar_launch_state_norms", "caar_layout_from_f90",
        "caar_layout_to_f90", "caar_arrays_alloc", "caar_arrays_free", "caar_create", "caar_destroy", "caar_upload",
        "caar_download", "caar_upload_f90", "caar_download_f90", "caar_upload_f90_arrays", "caar_run", "caar_run_steps",
        "caar_sync", "caar_device_arrays", "caar_stream", "caar_state_norms", "caar_map_host", "caar_run_mapped",
        "caar_unmap_host")
    # include/caar_tuning.h: variants, cache policy, placement, measurement, debug — not part of the boundary
    TUNING_SYMBOLS = (
        "caar_kernel_name", "caar_num_variants", "caar_select_variant", "caar_selected_variant", "caar_variant_info",
        "caar_set_xcd_chunked", "caar_set_cache_window", "caar_get_cache_window", "caar_set_adaptive_window",
        "caar_get_adaptive_window", "caar_adaptive_window_state", "caar_adaptive_window_reset", "caar_adaptive_window_lock_count",
        "caar_context_cache_window",
        "caar_set_fused_steps", "caar_get_fused_steps", "caar_has_fused_steps", "caar_arrays_alloc_ex",
        "caar_arrays_placement", "caar_create_ex", "caar_stream_copy", "caar_stream_copy_tuned",
        "caar_stream_copy_tuned_variants", "caar_stream_copy_tuned_info", "caar_traffic_skeleton", "caar_time_runs",
        "caar_reciprocal", "caar_debug_dp3d_violations")
    SYMBOLS = BOUNDARY_SYMBOLS + TUNING_SYMBOLS

    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise CaarError(
                "HIP extension %s is missing: build it with `python -m tinman_sandbox_amd.build` "
                "(there is no CPU fallback)" % path)
        L = self.lib = C.CDLL(path)
        for s in self.SYMBOLS:
            getattr(L, s)  # AttributeError if the library does not export it
        vp = C.c_void_p
        L.caar_supported.argtypes = [C.c_int, C.c_int]
        L.caar_supported_ex.argtypes = [C.c_int, C.c_int, C.c_int]
        L.caar_debug_dp3d_violations.argtypes = [C.c_int]
        L.caar_debug_dp3d_violations.restype = C.c_longlong
        L.caar_strerror.argtypes = [C.c_int]
        L.caar_strerror.restype = C.c_char_p
        L.caar_array_len.argtypes = [C.POINTER(_CaarDims), C.c_int]
        L.caar_array_len.restype = C.c_longlong
        L.caar_algorithmic_bytes.argtypes = [C.c_int, C.c_int, C.c_int]
        L.caar_algorithmic_bytes.restype = C.c_longlong
        L.caar_launch.argtypes = [C.POINTER(_CaarDims), C.POINTER(_CaarArrays), vp,
                                  C.POINTER(_CaarParams), vp]
        L.caar_launch_steps.argtypes = [C.POINTER(_CaarDims), C.POINTER(_CaarArrays), vp, C.POINTER(_CaarParams), C.c_int,
                                        C.c_int, vp]
        L.caar_launch_state_norms.argtypes = [C.POINTER(_CaarDims), C.POINTER(_CaarArrays), C.c_int,
                                              C.c_int, C.c_int, vp, vp]
        L.caar_sphere_operator.argtypes = [C.POINTER(_CaarDims), C.POINTER(_CaarArrays), vp, C.c_int, C.c_int,
                                           C.c_int, vp, vp, C.c_double, vp]
        L.caar_sphere_operator_range.argtypes = [C.POINTER(_CaarDims), C.POINTER(_CaarArrays), vp, C.c_int, C.c_int,
                                                 C.c_int, C.c_int, vp, vp, C.c_double, vp]
        L.caar_euler_step.argtypes = [C.POINTER(_CaarDims), C.POINTER(_CaarOperatorGeometry), vp, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_double, C.c_double, vp, vp, vp, vp]
        L.caar_sphere_operator_ex.argtypes = [C.POINTER(_CaarDims), C.POINTER(_CaarOperatorGeometry), vp, C.c_int, C.c_int,
                                              C.c_int, C.c_int, vp, vp, C.POINTER(_CaarOperatorScalars), vp]
        L.caar_preq_hydrostatic.argtypes = [C.POINTER(_CaarDims), C.c_int, vp, vp, vp, vp, C.c_double, vp, vp]
        L.caar_preq_omega_ps.argtypes = [C.POINTER(_CaarDims), C.c_int, vp, vp, vp, vp, vp]
        L.caar_sphere_operator_host.argtypes = [C.POINTER(_CaarDims), C.POINTER(_CaarArrays), vp, C.c_int, C.c_int, vp, vp,
                                                C.c_double]
        L.caar_preq_hydrostatic_host.argtypes = [C.POINTER(_CaarDims), vp, vp, vp, vp, C.c_double, vp]
        L.caar_preq_omega_ps_host.argtypes = [C.POINTER(_CaarDims), vp, vp, vp, vp]
        L.caar_reciprocal.argtypes = [vp, vp, C.c_longlong, vp]
        L.caar_kernel_name.argtypes = [C.c_int, C.c_int]
        L.caar_kernel_name.restype = C.c_char_p
        L.caar_num_variants.argtypes = [C.c_int, C.c_int]
        L.caar_select_variant.argtypes = [C.c_int, C.c_int, C.c_int]
        L.caar_variant_info.argtypes = [C.c_int, C.c_int, C.c_int]
        L.caar_variant_info.restype = C.c_char_p
        L.caar_set_xcd_chunked.argtypes = [C.c_int]
        L.caar_set_cache_window.argtypes = [C.c_longlong]
        L.caar_get_cache_window.restype = C.c_longlong
        L.caar_adaptive_window_state.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
        L.caar_adaptive_window_lock_count.restype = C.c_longlong
        L.caar_context_cache_window.argtypes = [vp]
        L.caar_context_cache_window.restype = C.c_longlong
        L.caar_selected_variant.argtypes = [C.c_int, C.c_int]
        L.caar_layout_from_f90.argtypes = [C.POINTER(_CaarDims), C.POINTER(_CaarArrays),
                                           C.POINTER(_CaarArrays), C.c_int, C.c_int, vp]
        L.caar_layout_to_f90.argtypes = [C.POINTER(_CaarDims), C.POINTER(_CaarArrays),
                                         C.POINTER(_CaarArrays), C.c_int, C.c_int, C.c_int, vp]
        L.caar_stream_copy.argtypes = [vp, vp, C.c_longlong, C.c_int, vp]
        L.caar_stream_copy_tuned.argtypes = [vp, vp, C.c_longlong, C.c_int, vp]
        L.caar_stream_copy_tuned_info.argtypes = [C.c_int]
        L.caar_stream_copy_tuned_info.restype = C.c_char_p
        L.caar_traffic_skeleton.argtypes = [C.POINTER(_CaarDims), C.POINTER(_CaarArrays),
                                            C.POINTER(_CaarParams), C.c_int, vp]
        L.caar_arrays_alloc.argtypes = [C.POINTER(vp), C.POINTER(_CaarDims), C.c_int, C.POINTER(_CaarArrays)]
        L.caar_arrays_alloc_ex.argtypes = [C.POINTER(vp), C.POINTER(_CaarDims), C.c_int, C.POINTER(_CaarPlacement),
                                           C.POINTER(_CaarArrays)]
        L.caar_arrays_free.argtypes = [vp]
        L.caar_arrays_placement.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        L.caar_create.argtypes = [C.POINTER(vp), C.POINTER(_CaarDims), C.c_int]
        L.caar_create_ex.argtypes = [C.POINTER(vp), C.POINTER(_CaarDims), C.c_int, C.POINTER(_CaarPlacement)]
        L.caar_destroy.argtypes = [vp]
        L.caar_destroy.restype = None
        L.caar_upload.argtypes = [vp, C.POINTER(_CaarArrays), C.c_int, C.c_int]
        L.caar_download.argtypes = [vp, C.POINTER(_CaarArrays), C.c_int, C.c_int, C.c_int]
        L.caar_upload_f90.argtypes = [vp, C.POINTER(_CaarArrays), C.c_int, C.c_int]
        L.caar_upload_f90_arrays.argtypes = [vp, C.POINTER(_CaarArrays), C.c_int, C.c_int, C.c_uint]
        L.caar_download_f90.argtypes = [vp, C.POINTER(_CaarArrays), C.c_int, C.c_int, C.c_int]
        L.caar_run.argtypes = [vp, C.POINTER(_CaarParams)]
        L.caar_sync.argtypes = [vp]
        L.caar_device_arrays.argtypes = [vp, C.POINTER(_CaarArrays)]
        L.caar_stream.argtypes = [vp]
        L.caar_stream.restype = vp
        L.caar_state_norms.argtypes = [vp, C.c_int, C.c_int, C.c_int, _dp]
        L.caar_time_runs.argtypes = [vp, C.POINTER(_CaarParams), C.c_int, C.POINTER(C.c_float)]
        L.caar_run_steps.argtypes = [vp, C.POINTER(_CaarParams), C.c_int, C.c_int]
        L.caar_set_fused_steps.argtypes = [C.c_int]
        L.caar_has_fused_steps.argtypes = [C.c_int, C.c_int, C.c_int]
        L.caar_map_host.argtypes = [C.POINTER(vp), C.POINTER(_CaarDims), C.POINTER(_CaarArrays), C.c_int]
        L.caar_run_mapped.argtypes = [vp, C.POINTER(_CaarParams)]
        L.caar_unmap_host.argtypes = [vp]

    def check(self, rc, what):
        if rc != 0:
            hint = ""
            if rc == -2 and os.path.exists(os.path.join(_HERE, "csrc", "libcaar_hip_extra.so")) and "extra" not in LIB_PATH:
                hint = ("; libcaar_hip_extra.so (CAAR_LIBRARY=extra) holds more: the Eulerian form beyond 128 levels, launch "
                        "shapes specialised for NLEV 26, 30, 32, 60, 64, 80, 96")
            raise CaarError("%s failed: rc=%d (%s)%s" % (what, rc, self.lib.caar_strerror(rc).decode(), hint))


_LIB = None


def library():
    global _LIB
    if _LIB is None:
        _LIB = CaarLibrary()
    return _LIB


# ----------------------------------------------------------------------------- layout
def array_shapes(np_, nlev, qsize_d, timelevels, num_elems):
    """Element-major layouts of Arrays::init_data, data_structures.cpp:14-31."""
    ne = num_elems
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


def algorithmic_bytes(np_, nlev, dry=False):
    """bytes one element update must move (inputs once, outputs once, RMW twice)."""
    pp = np_ * np_
    return 8 * ((20 if dry else 21) * pp * nlev + 2 * pp * (nlev + 1) + 13 * pp)


def shard_range(num_elems, rank, world_size):
    """Contiguous element slab [nets, nete) of `rank`: ceil(E/G) elements
    per rank, the last ranks may get fewer (or none)."""
    per = -(-num_elems // world_size)
    nets = min(rank * per, num_elems)
    return nets, min(nets + per, num_elems)


class _Arena:
    """Device memory from caar_arrays_alloc (placed for bandwidth, DESIGN.md section 5 "Placement"), freed when the last
    tensor that views it is gone."""

    def __init__(self, dims, device_index, place=None):
        self.lib = library()
        self.handle = C.c_void_p()
        self.ptrs = _CaarArrays()
        self.lib.check(self.lib.lib.caar_arrays_alloc_ex(C.byref(self.handle), C.byref(dims), device_index,
                                                         C.byref(place) if place is not None else None,
                                                         C.byref(self.ptrs)), "caar_arrays_alloc_ex")

    def spread(self):
        return self.lib.lib.caar_arrays_placement(self.handle, None, None) == 1

    def pool_bytes(self):
        """Size of the temporary pool the chunks were sampled from (0 for plain allocations)."""
        n, b = C.c_longlong(0), C.c_longlong(0)
        self.lib.lib.caar_arrays_placement(self.handle, C.byref(n), C.byref(b))
        return n.value * b.value

    def __del__(self):
        if getattr(self, "handle", None):
            self.lib.lib.caar_arrays_free(self.handle)
            self.handle = None


class _ArenaView:
    """One array of an _Arena through the CUDA array interface (torch.as_tensor shares the memory and keeps this object,
    hence the arena, alive)."""

    def __init__(self, arena, ptr, shape):
        self.arena = arena
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False), "version": 2,
                                         "strides": None}


class ElementArrays:
    """The 16 element arrays (Homme::Arrays) as torch float64 tensors on one device."""

    def __init__(self, np_, nlev, num_elems, qsize_d=1, timelevels=3, device="cpu", tensors=None, place=None):
        """`place`: None (the library's default placement), a CaarPlacement from placement(...), or "torch" for torch's
        own sixteen allocations (CAAR_PLACEMENT=torch in the environment does the same when `place` is None)."""
        self.np, self.nlev, self.num_elems = np_, nlev, num_elems
        self.qsize_d, self.timelevels = qsize_d, timelevels
        self.device = torch.device(device)
        shapes = array_shapes(np_, nlev, qsize_d, timelevels, num_elems)
        self.arena = None
        if place is None and os.environ.get("CAAR_PLACEMENT", "") == "torch":
            place = "torch"
        if tensors is None and self.device.type == "cuda" and not (isinstance(place, str) and place == "torch"):
            # the library's allocator: every array spread over the device's address classes (include/caar.h
            # caar_arrays_alloc_ex)
            idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
            with torch.cuda.device(idx):
                self.arena = _Arena(_CaarDims(np_, nlev, qsize_d, timelevels, num_elems), idx, place)
                ptrs = [C.cast(getattr(self.arena.ptrs, f), C.c_void_p).value for f, _ in _CaarArrays._fields_]
                tensors = {}
                for n, p in zip(ARRAY_NAMES, ptrs):
                    t = torch.as_tensor(_ArenaView(self.arena, p, shapes[n]), device=torch.device("cuda", idx))
                    tensors[n] = t.zero_()
        if tensors is None:
            tensors = {n: torch.zeros(s, dtype=torch.float64, device=self.device) for n, s in shapes.items()}
        for n in ARRAY_NAMES:
            t = tensors[n]
            assert tuple(t.shape) == shapes[n] and t.dtype == torch.float64 and t.is_contiguous(), n
        self.t = tensors

    def __getitem__(self, name):
        return self.t[name]

    @classmethod
    def from_numpy(cls, arrs, device="cpu"):
        ne, tl, nlev, np_, _ = arrs["elem_state_dp3d"].shape
        qd = arrs["elem_state_Qdp"].shape[1]
        tens = {n: torch.from_numpy(np.ascontiguousarray(arrs[n])).to(device) for n in ARRAY_NAMES}
        return cls(np_, nlev, ne, qd, tl, device, tens)

    def to_numpy(self):
        return {n: self.t[n].detach().cpu().numpy().copy() for n in ARRAY_NAMES}

    def to(self, device):
        return ElementArrays(self.np, self.nlev, self.num_elems, self.qsize_d, self.timelevels, device,
                             {n: self.t[n].to(device) for n in ARRAY_NAMES})

    def clone(self):
        return ElementArrays(self.np, self.nlev, self.num_elems, self.qsize_d, self.timelevels,
                             self.device, {n: self.t[n].clone() for n in ARRAY_NAMES})

    def nbytes(self):
        return sum(t.numel() * 8 for t in self.t.values())

    def dims(self):
        return _CaarDims(self.np, self.nlev, self.qsize_d, self.timelevels, self.num_elems)

    def pointers(self):
        return _CaarArrays(*[C.cast(self.t[n].data_ptr(), _dp) for n in ARRAY_NAMES])

    def init_data(self, first_elem=0):
        """The reference's closed-form initialiser (Arrays::init_data, data_structures.cpp:42-92
        == fortran/main.F90:103-154).  `first_elem` is the global index of local element 0,
        so that a rank holding the slab [nets, nete) builds exactly its part of the global
        arrays.  The three transcendental tables are formed with libm (math.sin/cos), like
        the reference, then broadcast on the device."""
        np_, nlev, ne, tl = self.np, self.nlev, self.num_elems, self.timelevels
        dev, f64 = self.device, torch.float64
        ip = torch.arange(1, np_ + 1, dtype=f64, device=dev).view(np_, 1)   # iip
        jp = torch.arange(1, np_ + 1, dtype=f64, device=dev).view(1, np_)   # jjp
        il = torch.arange(1, nlev + 1, dtype=f64, device=dev).view(nlev, 1, 1)
        iie = (torch.arange(ne, dtype=f64, device=dev) + (first_elem + 1))
        fcor = torch.tensor([[math.sin(i + j) for j in range(1, np_ + 1)] for i in range(1, np_ + 1)],
                            dtype=f64, device=dev)
        phi0 = torch.tensor([[math.cos(i + 3 * j) for j in range(1, np_ + 1)] for i in range(1, np_ + 1)],
                            dtype=f64, device=dev)
        qdp0 = torch.tensor([[[1.0 + math.sin(float(i * j * l)) for j in range(1, np_ + 1)]
                              for i in range(1, np_ + 1)] for l in range(1, nlev + 1)], dtype=f64, device=dev)
        t = self.t
        t["elem_fcor"][:] = fcor
        t["elem_metdet"][:] = ip * jp
        t["elem_rmetdet"][:] = 1.0 / (ip * jp)
        t["elem_spheremp"][:] = (2 * ip).expand(np_, np_)
        t["elem_state_phis"][:] = ip + jp
        t["elem_D"].zero_()
        t["elem_D"][..., 0, 0] = 1.0
        t["elem_D"][..., 1, 1] = 2.0
        t["elem_Dinv"].zero_()
        t["elem_Dinv"][..., 0, 0] = 1.0
        t["elem_Dinv"][..., 1, 1] = 0.5
        t["elem_derived_phi"][:] = phi0 + il
        t["elem_derived_vn0"].fill_(1.0)
        t["elem_derived_pecnd"].fill_(1.0)
        t["elem_derived_omega_p"][:] = (jp * jp).expand(np_, np_)
        t["elem_derived_eta_dot_dpdn"].zero_()
        t["elem_state_Qdp"].zero_()
        t["elem_state_Qdp"][:, 0, 0] = qdp0
        e = iie.view(ne, 1, 1, 1, 1)
        it = torch.arange(1, tl + 1, dtype=f64, device=dev).view(1, tl, 1, 1, 1)
        l5, i5, j5 = il.view(1, 1, nlev, 1, 1), ip.view(1, 1, 1, np_, 1), jp.view(1, 1, 1, 1, np_)
        # same operand order as data_structures.cpp:82-86
        t["elem_state_dp3d"][:] = 10.0 * l5 + e + i5 + j5 + it
        base = 1.0 + 0.5 * l5 + i5 + j5 + 0.2 * e
        t["elem_state_v"][..., 0] = base + 2.0 * it
        t["elem_state_v"][..., 1] = base + 3.0 * it
        t["elem_state_T"][:] = 1000.0 - l5 - i5 - j5 + 0.1 * e + it
        return self


# ------------------------------------------------- reference structs (data_structures.hpp)
class Constants:
    """Constants::init_data, data_structures.cpp:117-127."""

    def __init__(self):
        self.Rwater_vapor = 461.5
        self.Rgas = 287.04
        self.cp = 1005.0
        self.kappa = self.Rgas / self.cp
        self.rrearth = 1.0 / 6.376e6
        self.eta_ave_w = 1.0


class Control:
    """Control::init_data, data_structures.cpp:129-139."""

    def __init__(self, num_elems):
        self.nets, self.nete = 0, num_elems
        self.n0, self.np1, self.nm1, self.qn0 = 0, 1, 2, 0
        self.dt2 = 1.0
        # level_vectorized_ppscan/Control.hpp:48-49.  > 0: vertically Lagrangian (all the
        # reference's built variants); 0: Eulerian, needs HVCoord.hybi (parity unpinned)
        self.rsplit = 1


class HVCoord:
    """HVCoord::init_data, data_structures.cpp:141-149."""

    def __init__(self, nlev):
        self.ps0 = 10.0
        self.hyai = np.array([nlev + 1 - i for i in range(nlev + 1)], dtype=np.float64)
        self.hybi = None  # nlev+1 interface coefficients (hybvcoord_mod.F90:19); read only when rsplit == 0


NP4_DVV_VALUES = (
    -3.0000000000000000, -0.80901699437494745, 0.30901699437494745, -0.50000000000000000,
    4.0450849718747373, 0.00000000000000000, -1.11803398874989490, 1.54508497187473700,
    -1.5450849718747370, 1.11803398874989490, 0.00000000000000000, -4.04508497187473730,
    0.5000000000000000, -0.30901699437494745, 0.80901699437494745, 3.000000000000000000)


def gll_derivative_matrix(np_):
    """Dvv[i][j] = l_j'(x_i) on the np_-point Gauss-Lobatto-Legendre grid.  The reference
    hard-codes the np=4 values only (Derivative::init_data, data_structures.cpp:152-162,
    which this reproduces to rounding); np=8 needs a generated matrix."""
    N = np_ - 1
    x = -np.cos(np.pi * np.arange(np_) / N)
    P = np.polynomial.legendre.Legendre.basis(N)
    dP, d2P = P.deriv(), P.deriv(2)
    for _ in range(100):  # Newton on P_N'(x) for the interior nodes
        dx = dP(x[1:-1]) / d2P(x[1:-1])
        x[1:-1] -= dx
        if np.max(np.abs(dx)) < 1e-16:
            break
    x[0], x[-1] = -1.0, 1.0
    L = P(x)
    D = np.zeros((np_, np_))
    for i in range(np_):
        for j in range(np_):
            if i != j:
                D[i, j] = L[i] / (L[j] * (x[i] - x[j]))
    D[0, 0] = -0.25 * N * (N + 1)
    D[N, N] = 0.25 * N * (N + 1)
    return D


class Derivative:
    """Derivative::init_data, data_structures.cpp:151-163 (np=4 literals), GLL otherwise."""

    def __init__(self, np_, f32_rounded=False):
        if np_ == 4:
            vals = np.array(NP4_DVV_VALUES, dtype=np.float64)
            if f32_rounded:  # the Fortran driver's single-precision literals, main.F90:83-96
                vals = vals.astype(np.float32).astype(np.float64)
            self.Dvv = np.ascontiguousarray(vals.reshape(4, 4).T)  # Dvv[i][j] = values[j*np + i]
        else:
            self.Dvv = gll_derivative_matrix(np_)


class TestData:
    """Homme::TestData, data_structures.hpp:78-89."""
    __test__ = False  # not a pytest class

    def __init__(self):
        self.arrays = None
        self.constants = None
        self.control = None
        self.deriv = None
        self.hvcoord = None
        self._dvv_dev = None
        self._dvv_key = None
        self._hybi_key = None

    def init_data(self, num_elems, np_=4, nlev=72, device="cuda", first_elem=0, place=None):
        """TestData::init_data, data_structures.cpp:165-172.  `place`: see ElementArrays."""
        self.arrays = ElementArrays(np_, nlev, num_elems, device=device, place=place).init_data(first_elem)
        self.constants = Constants()
        self.control = Control(num_elems)
        self.hvcoord = HVCoord(nlev)
        self.deriv = Derivative(np_)
        return self

    @classmethod
    def from_numpy(cls, arrs, Dvv, scalars, device="cuda"):
        """Wrap host arrays + the flat scalar dict used by the tests/oracle."""
        d = cls()
        d.arrays = ElementArrays.from_numpy(arrs, device)
        d.constants = Constants()
        for k in ("Rwater_vapor", "Rgas", "kappa", "rrearth", "eta_ave_w"):
            setattr(d.constants, k, float(scalars[k]))
        d.control = Control(d.arrays.num_elems)
        for k in ("nets", "n0", "np1", "nm1", "qn0"):
            setattr(d.control, k, int(scalars[k]))
        if scalars.get("nete") is not None:
            d.control.nete = int(scalars["nete"])
        d.control.dt2 = float(scalars["dt2"])
        d.hvcoord = HVCoord(d.arrays.nlev)
        d.hvcoord.ps0 = float(scalars["ps0"])
        d.hvcoord.hyai = np.ascontiguousarray(scalars["hyai"], dtype=np.float64)
        d.control.rsplit = int(scalars.get("rsplit", 1))
        if scalars.get("hybi") is not None:
            d.hvcoord.hybi = np.ascontiguousarray(scalars["hybi"], dtype=np.float64)
        d.deriv = Derivative(d.arrays.np)
        d.deriv.Dvv = np.ascontiguousarray(Dvv, dtype=np.float64)
        return d

    def update_time_levels(self):
        """TestData::update_time_levels, data_structures.cpp:174-180."""
        c = self.control
        c.np1, c.nm1, c.n0 = c.nm1, c.n0, c.np1

    def dvv_device(self):
        key = (self.deriv.Dvv.tobytes(), str(self.arrays.device))
        if self._dvv_key != key:
            self._dvv_dev = torch.from_numpy(np.ascontiguousarray(self.deriv.Dvv)).to(self.arrays.device)
            self._dvv_key = key
        return self._dvv_dev

    def hybi_device(self):
        h = np.ascontiguousarray(self.hvcoord.hybi, dtype=np.float64)
        if h.size != self.arrays.nlev + 1:
            raise CaarError("HVCoord.hybi must hold nlev+1 values")
        key = (h.tobytes(), str(self.arrays.device))
        if self._hybi_key != key:
            self._hybi_dev = torch.from_numpy(h).to(self.arrays.device)
            self._hybi_key = key
        return self._hybi_dev

    def params(self, device_constants=False):
        """CaarParams for the C ABI; device_constants: also the device copy of hybi that the
        stateless caar_launch reads when rsplit == 0."""
        c, k, h = self.control, self.constants, self.hvcoord
        self._dvv_host = np.ascontiguousarray(self.deriv.Dvv, dtype=np.float64)
        hybi = hybi_dev = None
        if c.rsplit == 0:
            if h.hybi is None:
                raise CaarError("rsplit == 0 needs HVCoord.hybi")
            self._hybi_host = np.ascontiguousarray(h.hybi, dtype=np.float64)
            hybi = self._hybi_host.ctypes.data_as(_dp)
            if device_constants:
                hybi_dev = C.cast(C.c_void_p(self.hybi_device().data_ptr()), _dp)
        return _CaarParams(c.nets, c.nete, c.n0, c.np1, c.nm1, c.qn0, c.dt2, k.rrearth, k.eta_ave_w,
                           k.Rwater_vapor, k.Rgas, k.kappa, h.ps0, float(h.hyai[0]),
                           self._dvv_host.ctypes.data_as(_dp), c.rsplit, hybi, hybi_dev)


def _require_gpu(arrays):
    if arrays.device.type != "cuda":
        raise CaarError("compute_and_apply_rhs runs on the MI355X only: arrays live on %s "
                        "(there is no CPU fallback)" % arrays.device)


def compute_and_apply_rhs(data, stream=None):
    """Homme::compute_and_apply_rhs(TestData&), compute_and_apply_rhs.hpp:9 — one
    asynchronous launch on `stream` (default: torch's current stream) over the
    elements [control.nets, control.nete) of the device-resident arrays."""
    L = library()
    _require_gpu(data.arrays)
    if stream is None:
        stream = torch.cuda.current_stream(data.arrays.device)
    dims, ptrs, prm = data.arrays.dims(), data.arrays.pointers(), data.params(device_constants=True)
    dvv = data.dvv_device()
    with torch.cuda.device(data.arrays.device):  # the launch goes to the calling thread's current HIP device
        rc = L.lib.caar_launch(C.byref(dims), C.byref(ptrs), C.c_void_p(dvv.data_ptr()),
                               C.byref(prm), C.c_void_p(stream.cuda_stream))
    L.check(rc, "caar_launch")


def compute_and_apply_rhs_steps(data, nsteps, rotate=True, stream=None):
    """`nsteps` calls of compute_and_apply_rhs with TestData::update_time_levels between them (the reference's driver
    loop, main.cpp:113-121) — caar_launch_steps: one launch where a step-loop kernel exists.  Like the loop it replaces,
    it leaves data.control rotated nsteps times."""
    L = library()
    _require_gpu(data.arrays)
    if stream is None:
        stream = torch.cuda.current_stream(data.arrays.device)
    dims, ptrs, prm = data.arrays.dims(), data.arrays.pointers(), data.params(device_constants=True)
    dvv = data.dvv_device()
    with torch.cuda.device(data.arrays.device):
        rc = L.lib.caar_launch_steps(C.byref(dims), C.byref(ptrs), C.c_void_p(dvv.data_ptr()), C.byref(prm), nsteps,
                                     1 if rotate else 0, C.c_void_p(stream.cuda_stream))
    L.check(rc, "caar_launch_steps")
    if rotate:
        for _ in range(nsteps):
            data.update_time_levels()


def sphere_operator(which, field, data, ielem):
    """gradient_sphere (which=0) / divergence_sphere (1) / vorticity_sphere (2) of the
    reference (sphere_operators.hpp:9-16) applied to a batch of levels of element `ielem`:
    `field` is a device tensor [nlevels][np][np] (gradient) or [nlevels][np][np][2]."""
    L = library()
    _require_gpu(data.arrays)
    np_ = data.arrays.np
    nl = field.shape[0]
    f = field.contiguous()
    want = (nl, np_, np_) if which == 0 else (nl, np_, np_, 2)
    assert tuple(f.shape) == want and f.dtype == torch.float64
    out = torch.empty((nl, np_, np_, 2) if which == 0 else (nl, np_, np_), dtype=torch.float64, device=f.device)
    stream = torch.cuda.current_stream(data.arrays.device)
    dims, ptrs = data.arrays.dims(), data.arrays.pointers()
    L.check(L.lib.caar_sphere_operator(C.byref(dims), C.byref(ptrs), C.c_void_p(data.dvv_device().data_ptr()), which,
                                       ielem, nl, C.c_void_p(f.data_ptr()), C.c_void_p(out.data_ptr()),
                                       data.constants.rrearth, C.c_void_p(stream.cuda_stream)), "caar_sphere_operator")
    return out


def sphere_operator_all(which, field, data, e0=0, e1=None):
    """The same operator on elements [e0, e1) in one launch: `field` is a device tensor
    [e1-e0][nlevels][np][np] (gradient) or [e1-e0][nlevels][np][np][2]."""
    L = library()
    _require_gpu(data.arrays)
    np_ = data.arrays.np
    e1 = data.arrays.num_elems if e1 is None else e1
    f = field.contiguous()
    ne, nl = f.shape[0], f.shape[1]
    want = (e1 - e0, nl, np_, np_) if which == 0 else (e1 - e0, nl, np_, np_, 2)
    assert tuple(f.shape) == want and f.dtype == torch.float64
    out = torch.empty((ne, nl, np_, np_, 2) if which == 0 else (ne, nl, np_, np_), dtype=torch.float64, device=f.device)
    stream = torch.cuda.current_stream(data.arrays.device)
    dims, ptrs = data.arrays.dims(), data.arrays.pointers()
    L.check(L.lib.caar_sphere_operator_range(C.byref(dims), C.byref(ptrs), C.c_void_p(data.dvv_device().data_ptr()),
                                             which, e0, e1, nl, C.c_void_p(f.data_ptr()), C.c_void_p(out.data_ptr()),
                                             data.constants.rrearth, C.c_void_p(stream.cuda_stream)),
            "caar_sphere_operator_range")
    return out


def sphere_operator_ex(name, field, geometry, Dvv, rrearth, out=None, alpha=1.0, beta=0.0, nu_ratio=1.0, e0=0):
    """One of the sphere operators next to the CAAR path (caar_sphere_operator_ex; reference
    level_vectorized_ppscan/SphereOperators.hpp:271-993) on device tensors: `field` [ne][nlevels][np][np](,2) for the
    elements e0 .. e0+ne-1, `geometry` a dict of device tensors [num_elems][np][np](...) named as the members of
    CaarOperatorGeometry (only those the operator reads), `Dvv` a device tensor [np][np].  `out` is the tensor the
    *_update operators accumulate into (modified in place and returned)."""
    L = library()
    code, vin, vout = SPHERE_OPERATORS[name]
    f = field.contiguous()
    if not f.is_cuda or f.dtype != torch.float64:
        raise CaarError("sphere_operator_ex needs float64 device tensors (no CPU fallback)")
    ne, nl, np_ = f.shape[0], f.shape[1], f.shape[2]
    assert tuple(f.shape) == ((ne, nl, np_, np_, 2) if vin else (ne, nl, np_, np_))
    oshape = (ne, nl, np_, np_, 2) if vout else (ne, nl, np_, np_)
    if name == "laplace_tensor_replace":  # K:600-637: the input field is replaced by the result
        if f.data_ptr() != field.data_ptr():
            raise CaarError("laplace_tensor_replace works in place and needs a contiguous field")
        out = f
    if out is None:
        out = torch.empty(oshape, dtype=torch.float64, device=f.device)
    assert tuple(out.shape) == oshape and out.is_contiguous() and out.dtype == torch.float64
    g = _CaarOperatorGeometry()
    keep = []
    num_elems = None
    for n, _ in _CaarOperatorGeometry._fields_:
        t = geometry.get(n)
        if t is not None:
            t = t.contiguous()
            keep.append(t)
            num_elems = t.shape[0]
            setattr(g, n, t.data_ptr())
    dims = _CaarDims(np_, nl, 1, 1, num_elems if num_elems is not None else e0 + ne)
    sc = _CaarOperatorScalars(rrearth, alpha, beta, nu_ratio)
    dv = Dvv.contiguous()
    stream = torch.cuda.current_stream(f.device)
    L.check(L.lib.caar_sphere_operator_ex(C.byref(dims), C.byref(g), C.c_void_p(dv.data_ptr()), code, e0, e0 + ne, nl,
                                          C.c_void_p(f.data_ptr()), C.c_void_p(out.data_ptr()), C.byref(sc),
                                          C.c_void_p(stream.cuda_stream)), "caar_sphere_operator_ex")
    return out


def euler_step(vstar, Qdp, geometry, Dvv, qsize, qn0, dt, rrearth, e0=0, out=None):
    """The tracer step of the reference's EulerStepFunctor.hpp:32-68 (caar_euler_step): qtens(q) = Qdp(qn0, q) -
    dt * divergence_sphere(vstar * Qdp(qn0, q)) for q < qsize.  Device tensors: `vstar` [ne][nlev][np][np][2] for the
    elements e0 .. e0+ne-1, `Qdp` the whole [num_elems][qsize_d][2][nlev][np][np] array, `geometry` a dict with Dinv,
    metdet, rmetdet [num_elems][np][np](...), `Dvv` [np][np].  Returns qtens [ne][qsize][nlev][np][np]."""
    L = library()
    v, q = vstar.contiguous(), Qdp.contiguous()
    if not v.is_cuda or v.dtype != torch.float64 or not q.is_cuda or q.dtype != torch.float64:
        raise CaarError("euler_step needs float64 device tensors (no CPU fallback)")
    ne, nl, np_ = v.shape[0], v.shape[1], v.shape[2]
    assert tuple(v.shape) == (ne, nl, np_, np_, 2)
    num_elems, qsize_d = q.shape[0], q.shape[1]
    assert tuple(q.shape) == (num_elems, qsize_d, 2, nl, np_, np_) and e0 + ne <= num_elems
    if out is None:
        out = torch.empty((ne, qsize, nl, np_, np_), dtype=torch.float64, device=v.device)
    assert tuple(out.shape) == (ne, qsize, nl, np_, np_) and out.is_contiguous() and out.dtype == torch.float64
    g = _CaarOperatorGeometry()
    keep = []
    for n in ("Dinv", "metdet", "rmetdet"):
        t = geometry[n].contiguous()
        assert t.shape[0] == num_elems
        keep.append(t)
        setattr(g, n, t.data_ptr())
    dims = _CaarDims(np_, nl, qsize_d, 1, num_elems)
    dv = Dvv.contiguous()
    stream = torch.cuda.current_stream(v.device)
    L.check(L.lib.caar_euler_step(C.byref(dims), C.byref(g), C.c_void_p(dv.data_ptr()), e0, e0 + ne, qsize, qn0, dt,
                                  rrearth, C.c_void_p(v.data_ptr()), C.c_void_p(q.data_ptr()),
                                  C.c_void_p(out.data_ptr()), C.c_void_p(stream.cuda_stream)), "caar_euler_step")
    return out


def state_norms(data, tl=None):
    """(||v||, ||T||, ||dp3d||) of time level `tl` (default np1) over [nets, nete):
    the arithmetic of print_results_2norm, P:372-399, evaluated on the device."""
    L = library()
    _require_gpu(data.arrays)
    c = data.control
    tl = c.np1 if tl is None else tl
    n = c.nete - c.nets
    out = torch.empty(max(n, 1) * 3, dtype=torch.float64, device=data.arrays.device)
    stream = torch.cuda.current_stream(data.arrays.device)
    dims, ptrs = data.arrays.dims(), data.arrays.pointers()
    rc = L.lib.caar_launch_state_norms(C.byref(dims), C.byref(ptrs), tl, c.nets, c.nete,
                                       C.c_void_p(out.data_ptr()), C.c_void_p(stream.cuda_stream))
    L.check(rc, "caar_launch_state_norms")
    per = out.cpu().numpy()[: 3 * n].reshape(n, 3)
    s = [0.0, 0.0, 0.0]
    for e in range(n):  # P:388-390: plain running sums in element order
        for f in range(3):
            s[f] += per[e, f]
    return tuple(math.sqrt(x) for x in s)


def print_results_2norm(data):
    """P:392-396 output format."""
    v, t, d = state_norms(data)
    print("   ---> Norms:\n          ||v||_2  = %.17g\n          ||T||_2  = %.17g\n"
          "          ||dp||_2 = %.17g" % (v, t, d))
    return v, t, d
