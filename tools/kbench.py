#!/usr/bin/env python3
"""Times every compiled kernel variant of one (np, nlev) on the GPU, plus the memory
ceilings (stream copy, traffic skeleton), interleaved A/B in one process.

    python tools/kbench.py [--np 4] [--nlev 72] [--elems 10000] [--reps 20] [--rounds 3]

Prints one line per variant: median kernel ms (HIP events on the launch stream),
element-updates/s, algorithmic GB/s and fraction of the 8 TB/s HBM peak, and checks
each variant's result against variant 0 (<=1e-12 scaled error).
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import tinman_sandbox_amd as tsa  # noqa: E402


def time_ms(fn, reps, stream):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--np", type=int, default=4, dest="np_")
    ap.add_argument("--nlev", type=int, default=72)
    ap.add_argument("--elems", type=int, default=10000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--variants", type=str, default="")
    ap.add_argument("--skeletons", type=int, default=1, help="how many traffic-skeleton variants to time")
    ap.add_argument("--json", type=str, default="")
    a = ap.parse_args()
    L = tsa.library()
    lib = L.lib
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    data = tsa.TestData().init_data(a.elems, a.np_, a.nlev, device=dev)
    balg = tsa.algorithmic_bytes(a.np_, a.nlev)
    nvar = lib.caar_num_variants(a.np_, a.nlev)
    which = [int(x) for x in a.variants.split(",")] if a.variants else list(range(nvar))
    results = {}

    # correctness of each variant vs variant 0 on a fresh copy
    ref = None
    for v in [0] + [x for x in which if x != 0]:
        d = tsa.TestData().init_data(64, a.np_, a.nlev, device=dev)
        L.check(lib.caar_select_variant(a.np_, a.nlev, v), "select")
        tsa.compute_and_apply_rhs(d)
        torch.cuda.synchronize()
        out = {n: d.arrays[n].clone() for n in tsa.caar.MUTATED}
        if ref is None:
            ref = out
        else:
            for n in out:
                err = float((out[n] - ref[n]).abs().max() / ref[n].abs().max().clamp_min(1e-300))
                assert err <= 1e-12, (v, n, err)

    def run_variant(v, chunked=-1):
        L.check(lib.caar_select_variant(a.np_, a.nlev, v), "select")
        lib.caar_set_xcd_chunked(chunked)
        t = time_ms(lambda: tsa.compute_and_apply_rhs(data, stream), a.reps, stream)
        lib.caar_set_xcd_chunked(-1)
        return t

    # memory ceilings
    n_copy = 1 << 28  # 2 GiB src + 2 GiB dst
    src = torch.empty(n_copy, dtype=torch.float64, device=dev).fill_(1.0)
    dst = torch.empty_like(src)

    def copy(lb):
        return time_ms(lambda: L.check(lib.caar_stream_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()),
                                                            n_copy, lb, C.c_void_p(stream.cuda_stream)), "copy"),
                       a.reps, stream)

    dims, ptrs, prm = data.arrays.dims(), data.arrays.pointers(), data.params()

    def skeleton(sv):
        return time_ms(lambda: L.check(lib.caar_traffic_skeleton(C.byref(dims), C.byref(ptrs), C.byref(prm), sv,
                                                                 C.c_void_p(stream.cuda_stream)), "skel"),
                       a.reps, stream)

    times = {("variant", v): [] for v in which}
    times[("roundrobin", which[0])] = []  # default variant with the XCD-chunked element mapping
    times[("copy", 8)] = []
    times[("copy", 16)] = []
    if a.np_ == 4 or (a.np_ == 8 and a.nlev == 72):
        for sv in range(min(a.skeletons, 7) if a.np_ == 8 else a.skeletons):
            times[("skeleton", sv)] = []
    for _ in range(a.rounds):
        for key in list(times):
            if key[0] == "variant":
                times[key].append(run_variant(key[1]))
            elif key[0] == "roundrobin":
                times[key].append(run_variant(key[1], 1))
                times.setdefault(("dealt", key[1]), []).append(run_variant(key[1], 0))
            elif key[0] == "copy":
                times[key].append(copy(key[1]))
            else:
                times[key].append(skeleton(key[1]))
    lib.caar_select_variant(a.np_, a.nlev, 0)

    def med(x):
        return sorted(x)[len(x) // 2]

    print("np=%d nlev=%d elems=%d  B_alg=%d B/elem  reps=%d rounds=%d" % (a.np_, a.nlev, a.elems, balg, a.reps, a.rounds))
    for key, ts in times.items():
        ms = med(ts)
        if key[0] == "variant":
            v = key[1]
            gbs = balg * a.elems / (ms * 1e-3) / 1e9
            name = lib.caar_variant_info(a.np_, a.nlev, v).decode()
            print("variant %d  %8.4f ms  %7.3f M upd/s  %7.1f GB/s alg  %5.1f%% of 8TB/s  [min %.4f max %.4f]  %s" % (
                v, ms, a.elems / ms / 1e3, gbs, gbs / 80.0, min(ts), max(ts), name))
            results["variant%d" % v] = dict(ms=ms, gbs=gbs, what=name)
        elif key[0] == "dealt":
            gbs = balg * a.elems / (ms * 1e-3) / 1e9
            print("variant %d elements dealt round-robin over the XCDs  %8.4f ms  %7.1f GB/s alg  %5.1f%% of 8TB/s" % (key[1], ms, gbs, gbs / 80.0))
        elif key[0] == "roundrobin":
            gbs = balg * a.elems / (ms * 1e-3) / 1e9
            print("variant %d XCD-chunked elements  %8.4f ms  %7.1f GB/s alg  %5.1f%% of 8TB/s" % (key[1], ms, gbs, gbs / 80.0))
            results["variant%d_xcd_chunked" % key[1]] = dict(ms=ms, gbs=gbs)
        elif key[0] == "copy":
            gbs = 2 * n_copy * 8 / (ms * 1e-3) / 1e9
            print("stream copy %2d B/lane  %8.4f ms  %7.1f GB/s (read+write)  %5.1f%% of 8TB/s" % (key[1], ms, gbs, gbs / 80.0))
            results["copy%d" % key[1]] = dict(ms=ms, gbs=gbs)
        else:
            gbs = balg * a.elems / (ms * 1e-3) / 1e9
            print("traffic skeleton %d    %8.4f ms  %7.1f GB/s alg  %5.1f%% of 8TB/s" % (key[1], ms, gbs, gbs / 80.0))
            results["skeleton%d" % key[1]] = dict(ms=ms, gbs=gbs)
    if a.json:
        json.dump(results, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
