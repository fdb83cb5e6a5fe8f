"""Fortran-ordered element arrays: what a HOMME/E3SM host holds.

A Fortran array v(np,np,2,nlev,timelevels) per element (element_state_mod.F90:17-23),
flattened over elements, is — seen as a C-ordered array — of shape
(ne, timelevels, nlev, 2, np, np) with the LAST axis the Fortran first index.  Logical
indices coincide with the C++ layout (C++ [a][b][c] == Fortran (a+1,b+1,c+1)),
so the conversion is an axis permutation; on the device it is done by
csrc/caar_layout.hip (caar_layout_from_f90 / caar_layout_to_f90), on host arrays by the
numpy views below (used for staging small data and as the definition the kernels are
tested against)."""
import ctypes as C

import numpy as np
import torch

from . import caar as _c

# axis permutation that turns the C++-layout array into the Fortran-ordered one
# (both viewed as C-ordered numpy arrays); its inverse goes back.
_TO_F90 = {
    "elem_D": (0, 4, 3, 2, 1),                  # [ie][a][b][r][c] -> (ie, c, r, b, a)
    "elem_Dinv": (0, 4, 3, 2, 1),
    "elem_fcor": (0, 2, 1),
    "elem_spheremp": (0, 2, 1),
    "elem_metdet": (0, 2, 1),
    "elem_rmetdet": (0, 2, 1),
    "elem_state_dp3d": (0, 1, 2, 4, 3),         # [ie][t][k][a][b] -> (ie, t, k, b, a)
    "elem_state_v": (0, 1, 2, 5, 4, 3),         # [ie][t][k][a][b][c] -> (ie, t, k, c, b, a)
    "elem_state_T": (0, 1, 2, 4, 3),
    "elem_state_phis": (0, 2, 1),
    "elem_state_Qdp": (0, 2, 1, 3, 5, 4),       # [ie][q][t][k][a][b] -> (ie, t, q, k, b, a)
    "elem_derived_eta_dot_dpdn": (0, 1, 3, 2),
    "elem_derived_omega_p": (0, 1, 3, 2),
    "elem_derived_phi": (0, 1, 3, 2),
    "elem_derived_pecnd": (0, 1, 3, 2),
    "elem_derived_vn0": (0, 1, 4, 3, 2),        # [ie][k][a][b][c] -> (ie, k, c, b, a)
}


def to_f90_numpy(arrs):
    """C++-layout dict of numpy arrays -> Fortran-ordered flat arrays (as C-ordered views' copies)."""
    return {n: np.ascontiguousarray(np.transpose(arrs[n], _TO_F90[n])) for n in _c.ARRAY_NAMES if n in arrs}


def from_f90_numpy(f90):
    out = {}
    for n in _c.ARRAY_NAMES:
        if n in f90:
            inv = np.argsort(_TO_F90[n])
            out[n] = np.ascontiguousarray(np.transpose(f90[n], inv))
    return out


def f90_shapes(np_, nlev, qsize_d, timelevels, ne):
    shapes = _c.array_shapes(np_, nlev, qsize_d, timelevels, ne)
    return {n: tuple(shapes[n][i] for i in _TO_F90[n]) for n in _c.ARRAY_NAMES}


class F90Arrays:
    """Device-resident Fortran-ordered flat arrays (torch float64), one per CaarArrays member."""

    def __init__(self, np_, nlev, num_elems, qsize_d=1, timelevels=3, device="cuda", tensors=None):
        self.np, self.nlev, self.num_elems, self.qsize_d, self.timelevels = np_, nlev, num_elems, qsize_d, timelevels
        self.device = torch.device(device)
        shapes = f90_shapes(np_, nlev, qsize_d, timelevels, num_elems)
        if tensors is None:
            tensors = {n: torch.zeros(s, dtype=torch.float64, device=self.device) for n, s in shapes.items()}
        for n in _c.ARRAY_NAMES:
            assert tuple(tensors[n].shape) == shapes[n] and tensors[n].is_contiguous(), n
        self.t = tensors

    @classmethod
    def from_numpy(cls, f90, np_, nlev, ne, qsize_d=1, timelevels=3, device="cuda"):
        return cls(np_, nlev, ne, qsize_d, timelevels, device,
                   {n: torch.from_numpy(np.ascontiguousarray(f90[n])).to(device) for n in _c.ARRAY_NAMES})

    def to_numpy(self):
        return {n: self.t[n].cpu().numpy().copy() for n in _c.ARRAY_NAMES}

    def pointers(self):
        return _c._CaarArrays(*[C.cast(self.t[n].data_ptr(), _c._dp) for n in _c.ARRAY_NAMES])


def ingest(f90, arrays, e0=0, e1=None, stream=None):
    """Fortran-ordered device arrays -> C++-layout ElementArrays (caar_layout_from_f90)."""
    L = _c.library()
    _c._require_gpu(arrays)
    e1 = arrays.num_elems if e1 is None else e1
    stream = stream or torch.cuda.current_stream(arrays.device)
    dims, src, dst = arrays.dims(), f90.pointers(), arrays.pointers()
    L.check(L.lib.caar_layout_from_f90(C.byref(dims), C.byref(src), C.byref(dst), e0, e1,
                                       C.c_void_p(stream.cuda_stream)), "caar_layout_from_f90")


def egress(arrays, f90, e0=0, e1=None, all_arrays=False, stream=None):
    """C++-layout ElementArrays -> Fortran-ordered device arrays (caar_layout_to_f90)."""
    L = _c.library()
    _c._require_gpu(arrays)
    e1 = arrays.num_elems if e1 is None else e1
    stream = stream or torch.cuda.current_stream(arrays.device)
    dims, src, dst = arrays.dims(), arrays.pointers(), f90.pointers()
    L.check(L.lib.caar_layout_to_f90(C.byref(dims), C.byref(src), C.byref(dst), e0, e1, int(all_arrays),
                                     C.c_void_p(stream.cuda_stream)), "caar_layout_to_f90")
