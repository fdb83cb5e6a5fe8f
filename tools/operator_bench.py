#!/usr/bin/env python3
"""Bandwidth of the stand-alone sphere operators over a whole element range: the three CAAR operators
(caar_sphere_operator_range) and the neighbouring ones (caar_sphere_operator_ex): bytes read + written
(field in + field out; second figure: with the per-element geometry the operator reads, once per workgroup) over the
HIP-event time.

    python tools/operator_bench.py                  # BASELINE element counts (40-100 us launches: fill/drain matters)
    python tools/operator_bench.py --scale 5        # 5x the elements: >= 1 GB per launch, fill/drain < 3 %

Logs: profiles/r02/operator_bench.log, profiles/r03/operator_bench_large.log"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinman_sandbox_amd as tsa  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        out = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, out


ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--only", default="", help="comma-separated operator names (default: all)")
a = ap.parse_args()
only = set(x for x in a.only.split(",") if x)
# doubles of per-element geometry per GLL point that each operator reads (caar_operators_ex.hip NEED_*)
GEO_DOUBLES = {"gradient_sphere": 4, "divergence_sphere": 6, "vorticity_sphere": 5, "divergence_sphere_wk": 5,
               "laplace_simple": 5, "laplace_tensor": 9, "laplace_tensor_replace": 9, "curl_sphere_wk_testcov": 5,
               "grad_sphere_wk_testcov": 10, "vlaplace_sphere_wk_contra": 16, "vlaplace_sphere_wk_cartesian": 15,
               "vlaplace_sphere_wk_cartesian_damped": 15, "gradient_sphere_update": 4, "divergence_sphere_update": 6}

for np_, nlev, E in ((4, 72, int(10000 * a.scale)), (4, 128, int(12500 * a.scale)), (8, 72, int(20000 * min(a.scale, 2.5)))):
    data = tsa.TestData().init_data(E, np_, nlev, device="cuda")
    s = data.arrays["elem_state_T"][:, 0].contiguous()
    v = data.arrays["elem_state_v"][:, 0].contiguous()
    print("np=%d nlev=%d E=%d" % (np_, nlev, E), flush=True)
    for which, name, f in ((0, "gradient_sphere", s), (1, "divergence_sphere", v), (2, "vorticity_sphere", v)):
        if only and name not in only:
            continue
        ms, out = timed(lambda: tsa.sphere_operator_all(which, f, data))
        byts = (f.numel() + out.numel()) * 8
        print("  %-40s %7.3f ms  %6.0f GB/s  (%6.0f with geometry, %.2f GB per launch)" % (
            name + " (range)", ms, byts / ms / 1e6, (byts + GEO_DOUBLES[name] * E * np_ * np_ * 8) / ms / 1e6, byts / 1e9), flush=True)
    A = data.arrays
    g = torch.Generator(device="cuda").manual_seed(1)

    def rnd(*shape):
        return torch.rand(shape, dtype=torch.float64, device="cuda", generator=g) + 0.5

    geo = {"D": A["elem_D"], "Dinv": A["elem_Dinv"], "metdet": A["elem_metdet"], "rmetdet": A["elem_rmetdet"],
           "spheremp": A["elem_spheremp"], "mp": rnd(E, np_, np_), "metinv": rnd(E, np_, np_, 2, 2),
           "tensorVisc": rnd(E, np_, np_, 2, 2), "vec_sph2cart": rnd(E, np_, np_, 3, 2)}
    dvv = data.dvv_device()
    for name, (code, vin, vout) in tsa.SPHERE_OPERATORS.items():
        if code < 3 or (only and name not in only):
            continue
        f = v if vin else s
        if name == "laplace_tensor_replace":
            f = s.clone()  # overwritten in place, launch after launch
        acc = torch.zeros((E, nlev, np_, np_) + ((2,) if vout else ()), dtype=torch.float64, device="cuda")
        upd = name.endswith("_update")
        ms, out = timed(lambda: tsa.sphere_operator_ex(name, f, geo, dvv, 1.5e-7, out=acc if upd else None))
        byts = (f.numel() + out.numel() * (2 if upd else 1)) * 8
        print("  %-40s %7.3f ms  %6.0f GB/s  (%6.0f with geometry, %.2f GB per launch)" % (
            name, ms, byts / ms / 1e6, (byts + GEO_DOUBLES[name] * E * np_ * np_ * 8) / ms / 1e6, byts / 1e9), flush=True)
    # the tracer step (caar_euler_step; EulerStepFunctor.hpp:32-68): 2 + 2 qsize field blocks per element
    for qsize in (() if only else (1, 4)):
        qdp = rnd(E, qsize, 2, nlev, np_, np_)
        qt = torch.empty((E, qsize, nlev, np_, np_), dtype=torch.float64, device="cuda")
        ms, out = timed(lambda: tsa.euler_step(v, qdp, geo, dvv, qsize, 0, 0.5, 1.5e-7, out=qt))
        byts = (v.numel() + 2 * out.numel()) * 8
        print("  %-40s %7.3f ms  %6.0f GB/s" % ("euler_step qsize=%d" % qsize, ms, byts / ms / 1e6), flush=True)
        del qdp, qt
    del data, geo
    torch.cuda.empty_cache()
