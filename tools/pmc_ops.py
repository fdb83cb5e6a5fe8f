#!/usr/bin/env python3
"""HBM-side traffic of the composite sphere operators (laplace_tensor, vlaplace_sphere_wk_cartesian) from the PMC counters,
at a batch that moves >= 1 GB per launch.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_fetch -- python3 tools/pmc_ops.py run
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_write -- python3 tools/pmc_ops.py run
    python3 tools/pmc_ops.py parse out_fetch out_write

Counters in separate passes, KiB units, calibrated on the 8 B/lane and 16 B/lane stream copies of the same run whose
byte counts are known (MI355X_MICROARCH.md, HBM / rocprofv3 section), as tools/pmc_parse.py does for the CAAR kernel."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
E, NP, NLEV = 50000, 4, 72
N_COPY = 1 << 27
OPS = {"laplace_tensor": 5, "vlaplace_sphere_wk_cartesian": 9, "divergence_sphere_wk": 3}
GEO = {"laplace_tensor": 9, "vlaplace_sphere_wk_cartesian": 15, "divergence_sphere_wk": 5}


def run():
    import ctypes as C
    import torch
    import tinman_sandbox_amd as tsa
    L = tsa.library()
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev)
    src = torch.ones(N_COPY, dtype=torch.float64, device=dev)
    dst = torch.empty_like(src)
    for lb in (8, 16):
        for _ in range(3):
            L.check(L.lib.caar_stream_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), N_COPY, lb,
                                           C.c_void_p(st.cuda_stream)), "copy")
        torch.cuda.synchronize()
    del src, dst
    g = torch.Generator(device="cuda").manual_seed(1)

    def rnd(*shape):
        return torch.rand(shape, dtype=torch.float64, device="cuda", generator=g) + 0.5

    geo = {"D": rnd(E, NP, NP, 2, 2), "Dinv": rnd(E, NP, NP, 2, 2), "metdet": rnd(E, NP, NP), "rmetdet": rnd(E, NP, NP),
           "spheremp": rnd(E, NP, NP), "mp": rnd(E, NP, NP), "metinv": rnd(E, NP, NP, 2, 2),
           "tensorVisc": rnd(E, NP, NP, 2, 2), "vec_sph2cart": rnd(E, NP, NP, 3, 2)}
    s = rnd(E, NLEV, NP, NP)
    v = rnd(E, NLEV, NP, NP, 2)
    dvv = rnd(NP, NP)
    for name in OPS:
        vin, vout = tsa.SPHERE_OPERATORS[name][1:]
        f = v if vin else s
        out = torch.empty((E, NLEV, NP, NP) + ((2,) if vout else ()), dtype=torch.float64, device="cuda")
        for _ in range(4):
            tsa.sphere_operator_ex(name, f, geo, dvv, 1.5e-7, out=out)
        torch.cuda.synchronize()
    print("pmc_ops run done")


def parse(fetch_dir, write_dir):
    def load(d, counter):
        rows = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == counter:
                    rows.append((r["Kernel_Name"], float(r["Counter_Value"]) * 1024))
        return rows

    def mean(xs):
        xs = list(xs)
        return sum(xs) / len(xs) if xs else float("nan")

    res = {"elements": E, "np": NP, "nlev": NLEV}
    raw = {}
    for counter, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        rows = load(d, counter)
        c8 = mean(v for k, v in rows if "stream_copy_kernel<double>" in k)
        c16 = mean(v for k, v in rows if "stream_copy_kernel" in k and "stream_copy_kernel<double>" not in k)
        raw[counter] = {"factor_8B_lane": N_COPY * 8 / c8, "factor_16B_lane": N_COPY * 8 / c16}
        for name, code in OPS.items():
            raw[counter][name] = mean(v for k, v in rows if "sphere_operator_ex_kernel<%d, %d>" % (NP, code) in k)
    res["calibration"] = raw
    blk = E * NLEV * NP * NP * 8
    for name in OPS:
        import tinman_sandbox_amd.caar as m
        vin, vout = m.SPHERE_OPERATORS[name][1:]
        # scalar fields move 8 B per lane, vector fields 16 B per lane: each counter is corrected with the matching copy
        rd = raw["FETCH_SIZE"][name] * raw["FETCH_SIZE"]["factor_16B_lane" if vin else "factor_8B_lane"]
        wr = raw["WRITE_SIZE"][name] * raw["WRITE_SIZE"]["factor_16B_lane" if vout else "factor_8B_lane"]
        alg_r = blk * (2 if vin else 1) + GEO[name] * E * NP * NP * 8
        alg_w = blk * (2 if vout else 1)
        res[name] = {"read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "algorithmic_read_bytes_incl_geometry": alg_r,
                     "algorithmic_write_bytes": alg_w, "traffic_over_algorithmic": (rd + wr) / (alg_r + alg_w)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        parse(sys.argv[2], sys.argv[3])
