#!/usr/bin/env python3
"""The adaptive cache window (include/caar.h) on a host whose call pattern changes: 10 000 elements NP=4 NLEV=72,
   phase A  the routine replayed back to back                     (the window pays: kept accumulators are found again)
   phase B  alternated with a default-policy kernel moving 768 MiB (whatever was kept is evicted: all-streaming is ahead)
   phase C  replayed again
each phase with the adaptive window, with the window forced (adaptive off) and all-streaming (window 0); CAAR time per call =
(sequence - neighbour alone).  Prints the policy the library has in force at the end of each adaptive phase.

    python tools/adaptive_window_demo.py [--elems 10000] [--calls 400]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinman_sandbox_amd as tsa  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--elems", type=int, default=10000)
ap.add_argument("--calls", type=int, default=400)
a = ap.parse_args()
lib = tsa.library().lib
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
data = tsa.TestData().init_data(a.elems, 4, 72, device=dev)
data.control.dt2, data.constants.eta_ave_w = 1.0e-6, 0.0   # timing only: thousands of calls stay finite
balg = tsa.algorithmic_bytes(4, 72) * a.elems
n_ev = 1 << 25
ev = [torch.ones(n_ev, dtype=torch.float64, device=dev) for _ in range(3)]
window = lib.caar_get_cache_window()


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        fn()
    e1.record(st)
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / n


def evict():
    torch.add(ev[0], ev[1], out=ev[2])


def caar():
    tsa.compute_and_apply_rhs(data, st)


def caar_then_evict():
    caar()
    evict()


def state():
    w, s, n = C.c_double(0), C.c_double(0), C.c_longlong(0)
    r = lib.caar_adaptive_window_state(C.c_void_p(data.arrays["elem_derived_vn0"].data_ptr()), C.byref(w), C.byref(s), C.byref(n))
    return "%s (last probe: window %.4f ms, all-streaming %.4f ms; %d probes)" % (
        {1: "window", 0: "all-streaming", -1: "unknown"}[r], w.value, s.value, n.value)


timed(caar, 100)
ev_ms = min(timed(evict, 20) for _ in range(2))
print("%d elements, %d calls per phase; neighbour alone %.4f ms" % (a.elems, a.calls, ev_ms))
for mode, adaptive, win, variant in (("adaptive", 1, window, 0), ("window forced", 0, window, 0), ("all-streaming forced", 0, 0, 0),
                                     ("nt twin (variant 1)", 0, 0, 1), ("all-streaming forced", 0, 0, 0), ("nt twin (variant 1)", 0, 0, 1)):
    lib.caar_select_variant(4, 72, variant)
    lib.caar_set_adaptive_window(adaptive)
    lib.caar_set_cache_window(win)
    lib.caar_adaptive_window_reset()
    for phase, fn, sub in (("A replay", caar, 0.0), ("B evicting neighbour", caar_then_evict, ev_ms), ("C replay", caar, 0.0)):
        whole = timed(fn, a.calls) - sub               # the whole phase, adaptation included
        tail = min(timed(fn, a.calls // 4) for _ in range(3)) - sub   # ... and its settled end (best of three: a block is
        #                                                                 now and then hit by a driver stall of tens of ms)
        print("%-22s %-22s CAAR %.4f ms per call over the phase (%.1f %% of peak), %.4f settled (%.1f %%)%s" % (
            mode, phase, whole, balg / whole / 8e7, tail, balg / tail / 8e7, "   policy: " + state() if adaptive else ""), flush=True)
lib.caar_select_variant(4, 72, 0)
lib.caar_set_adaptive_window(1)
lib.caar_set_cache_window(window)
