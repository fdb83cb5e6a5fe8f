#!/usr/bin/env python3
"""Benchmark of the hot path: element-RHS-updates/s of compute_and_apply_rhs on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for N > 1: under a launcher (WORLD_SIZE set) this process is one rank; started plainly with
--gpus N > 1 it becomes the launcher itself — before anything touches the GPU it starts N fresh rank processes
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set), relays rank 0's single JSON line
and exits with the ranks' exit code (self_launch below).

A "step" is one compute_and_apply_rhs call over every element resident on the GPU
(fp64, moist branch, the reference's closed-form synthetic arrays).  --gpus 1 is
BASELINE.json configs[1]: NP=4, NLEV=72, 10 000 elements.  --gpus N > 1 holds 12 500
elements per GPU, so --gpus 8 is configs[2] (100 000 elements sharded over 8 GPUs);
the same 12 500-element launch is also measured at N=1 ("other_configs") so the two
are comparable.  Elements shard embarrassingly: every rank owns a contiguous slab of
the global element range and there is no data-path collective (weak scaling).

Rank 0 prints ONE JSON line: metric/value (whole-job element updates per second),
"roofline" for the dominant kernel (algorithmic bytes per launch / HIP-event
kernel time, against the 8 TB/s HBM3E peak) and "cpu_baseline" (the reference's
own serial C++ path, or this repo's C port of it, on the host cores; N=1 only).
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--np", type=int, default=4, dest="np_")
    ap.add_argument("--nlev", type=int, default=72)
    ap.add_argument("--elems-per-gpu", type=int, default=None,
                    help="elements resident per GPU (weak scaling). Default: 10 000 at --gpus 1 (BASELINE.json "
                         "configs[1]); 12 500 at --gpus N > 1, so that --gpus 8 is configs[2] (100 000 elements "
                         "sharded over 8 GPUs)")
    ap.add_argument("--total-elems", type=int, default=None,
                    help="total elements of the job, cut into contiguous slabs of ceil(E/N) (overrides "
                         "--elems-per-gpu; the last rank may hold fewer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--isolated", action="store_true", help="(default since round 4; accepted for old command lines)")
    ap.add_argument("--no-isolated", action="store_true",
                    help="skip the per-launch timings after the timed region (roofline.kernel_ms_median/_min: max(K, 20) launches "
                         "back to back with an event after each; kernel_ms_isolated_*: 20 launches one by one)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short extra measurements of BASELINE.json configs[3] and [4]")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 --pmc child passes (the committed figure is used)")
    ap.add_argument("--no-steps-leg", action="store_true",
                    help="skip roofline.run_steps (the driver loop as one launch against single launches)")
    ap.add_argument("--no-interleaved", action="store_true",
                    help="skip the host-sequence measurements (roofline.interleaved: CAAR alternated with a tracer step / "
                         "a cache-evicting kernel, time levels rotating)")
    ap.add_argument("--no-dropin", action="store_true",
                    help="skip the `dropin` object (the reference driver's loop through Homme::compute_and_apply_rhs(TestData&) "
                         "on host-owned arrays, mapped and resident mode, in child processes)")
    ap.add_argument("--no-spinup", action="store_true",
                    help="skip the untimed device spin-up before the W warmup steps: the timed steps then run on a "
                         "device that is still ramping up (fresh process: clocks, TLBs, cache window; ~25 ms)")
    ap.add_argument("--cpu-seconds", type=float, default=1.5,
                    help="target wall seconds per host thread for the CPU baseline sample")
    return ap.parse_args()


def cgroup_cpu_quota():
    """CPUs' worth of run time the cgroup grants this job (cpu.max quota / period), or None if unlimited / unreadable."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return int(quota) / int(period)
    except (OSError, ValueError):
        pass
    return None


def host_cores():
    """Host threads the CPU baseline uses: the affinity mask, cut to the cgroup CPU
    quota when there is one and to the GPU box's per-GPU CPU share (16; override with
    CAAR_BENCH_CORES)."""
    n = len(os.sched_getaffinity(0))
    q = cgroup_cpu_quota()
    if q is not None:
        n = min(n, max(1, int(q)))
    return max(1, min(n, int(os.environ.get("CAAR_BENCH_CORES", "16"))))


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


REF_FLAGS = "g++ -std=c++11 -O3 -fPIC (the reference's own flags, cmake/SetCompilerFlags.cmake:56,123; oracle/Makefile REF_CXXFLAGS)"
PORT_FLAGS = "gcc -std=c99 -O2 -ffp-contract=off -fPIC (oracle/Makefile ORACLE_CFLAGS)"


def cpu_baseline(np_, nlev, seconds, elems=10000):
    """SURVEY 8(d) / BASELINE.md section 4: the CPU path on the GPU box's host cores, wall clock
    (time.perf_counter around the C call; the reference's own timer is CLOCK_THREAD_CPUTIME_ID, timer.cpp:37,
    which would hide stalls).  What is timed: oracle/_ref — the reference's own pointers_only C++ compiled from
    its sources by oracle/Makefile (kind "reference") — when it was built, else oracle/caar_oracle.c (kind "port").
      (a) ONE core, the whole `elems`-element data set (configs[1]: 10 000, out of cache), 1 warm-up + 3 timed calls;
      (b) all host cores, the same `elems` elements cut into one contiguous slab per thread, each thread with its own
          Control range (the reference's own sharding hook, data_structures.hpp:58-69), `seconds` of wall time each."""
    from oracle import pyoracle as po
    cores = host_cores()
    use_ref = po.have_reference(np_, nlev)
    O = po.Oracle()
    Dvv = O.dvv_np4(False) if np_ == 4 else O.dvv_gll(np_)
    sc = po.default_scalars(nlev)

    def make_runner(ne):
        arrs = O.init_arrays(np_, nlev, 1, 3, ne)
        if use_ref:
            R = po.Reference(np_, nlev)
            return lambda reps=1: R.compute_and_apply_rhs(arrs, Dvv, sc, reps)
        o = po.Oracle()
        return lambda reps=1: o.compute_and_apply_rhs(arrs, Dvv, sc, reps)

    # (a) one core, the full data set
    run = make_runner(elems)
    run()
    one = []
    for _ in range(3):
        t0 = time.perf_counter()
        run()
        one.append(time.perf_counter() - t0)
    del run
    one_med = sorted(one)[1]

    # (b) `cores` threads: slabs of the same data set
    def threaded(nthreads, total_updates=None):
        per_thread = -(-elems // nthreads)
        reps = max(3, int(round(seconds / (one_med * per_thread / elems))))
        if total_updates is not None:   # a bounded sample: the same number of element updates as the `cores` leg made
            reps = max(2, -(-int(total_updates) // (nthreads * per_thread)))
        runners = [make_runner(per_thread) for _ in range(nthreads)]
        for r in runners:
            r()  # warm-up call (touches the memory)
        barrier = threading.Barrier(nthreads + 1)

        def work(r):
            barrier.wait()
            r(reps)   # the reps calls are one C call (the reference driver's loop): no Python between them, the GIL released
            barrier.wait()

        ths = [threading.Thread(target=work, args=(r,)) for r in runners]
        for t in ths:
            t.start()
        barrier.wait()
        t0 = time.perf_counter()
        barrier.wait()
        wall = time.perf_counter() - t0
        for t in ths:
            t.join()
        return per_thread, reps, wall

    per_thread, reps, wall = threaded(cores)
    # (c) every hardware thread of the affinity mask (the whole box, not the per-GPU share), same protocol.  Where the job
    # runs under a cgroup CPU quota (the GPU pool's boxes: 16 CPUs' worth for a one-GPU job) the threads beyond the quota
    # only time-share it: the figure is then what THIS JOB may use of the box, stated as such.
    all_n = len(os.sched_getaffinity(0))
    quota = cgroup_cpu_quota()
    if all_n > cores:
        a_per, a_reps, a_wall = threaded(all_n, total_updates=cores * per_thread * reps)
        all_threads = {"value": all_n * a_per * a_reps / a_wall, "cores": all_n, "elements_per_thread": a_per, "calls": a_reps,
                       "wall_seconds": a_wall, "cgroup_cpu_quota": quota,
                       "note": ("%d threads time-sharing a cgroup quota of %.1f CPUs (oversubscribed and throttled: slower than the "
                                "%d-thread leg): not a whole-box figure, none can be measured from inside this job; the sample is "
                                "the same number of element updates as the %d-thread leg made" % (all_n, quota, cores, cores))
                               if quota is not None and quota < all_n else
                               "every hardware thread in the affinity mask; an 8-GPU node shares them between its 8 GPUs, so "
                               "the per-GPU comparison is `value` (%d threads)" % cores}
    else:
        all_threads = {"value": None, "cores": all_n, "cgroup_cpu_quota": quota,
                       "note": "the affinity mask holds no more threads than `cores`"}
    balg = 8 * (21 * np_ * np_ * nlev + 2 * np_ * np_ * (nlev + 1) + 13 * np_ * np_)
    what = "oracle/_ref (reference cxx/pointers_only)" if use_ref else "oracle/caar_oracle.c"
    value = cores * per_thread * reps / wall
    return {
        "value": value,
        "unit": "element-updates/s",
        "cores": cores,
        "kind": "reference" if use_ref else "port",
        "cpu_model": cpu_model(),
        "nproc": os.cpu_count(),
        "nproc_affinity": len(os.sched_getaffinity(0)),
        "flags": REF_FLAGS if use_ref else PORT_FLAGS,
        "clock": "wall (time.perf_counter around the calls)",
        "algorithmic_GBs": value * balg / 1e9,
        "single_core": {"value": elems / one_med, "elements": elems, "calls": 3, "warmup_calls": 1,
                        "seconds_per_call": one, "algorithmic_GBs": elems / one_med * balg / 1e9,
                        "min_seconds": min(one), "median_seconds": one_med},
        "single_core_value": elems / one_med,
        "all_threads": all_threads,
        "sample": "all cores: %d threads x %d elements (slabs of the %d-element data set) x %d calls of %s, NP=%d NLEV=%d, "
                  "wall %.2fs; one core: %d elements x 3 calls (median %.3fs)" % (
                      cores, per_thread, elems, reps, what, np_, nlev, wall, elems, one_med),
    }


def measure_config(np_, nlev, elems, steps, warmup):
    """One more configuration, measured by this same script in a FRESH process (the headline line of a child run with
    --no-other-configs): where the driver places the arrays moves the rate by 3-5 % (DESIGN.md section 5 "Placement"), and
    arrays allocated after the allocate/free traffic of a long-running process tend to land badly; a fresh process is what
    a host running that configuration would be."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--np", str(np_), "--nlev", str(nlev),
           "--elems-per-gpu", str(elems), "--steps", str(steps), "--warmup", str(warmup), "--no-other-configs",
           "--no-cpu-baseline", "--no-interleaved", "--no-live-traffic", "--no-dropin"]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not line:
        return {"workload": "NP=%d NLEV=%d num_elems=%d" % (np_, nlev, elems), "error": (r.stderr or r.stdout)[-400:]}
    j = json.loads(line[-1])
    roof = j["roofline"]
    return {"workload": "NP=%d NLEV=%d num_elems=%d" % (np_, nlev, elems), "kernel_ms": roof["kernel_ms"],
            "traffic": roof["traffic"], "traffic_source": roof.get("traffic_source", TRAFFIC_SOURCE),
            "element_updates_per_s": elems / (roof["kernel_ms"] * 1e-3), "achieved_GBs": roof["achieved"],
            "frac_of_hbm_peak": roof["frac"], "achieved_all_streaming_GBs": roof.get("achieved_all_streaming"),
            "algorithmic_bytes_per_element": roof["algorithmic_bytes_per_element"], "kernel": j["config"]["kernel"],
            "run_steps": roof.get("run_steps"),
            # the kernel against its own arithmetic-free traffic (same bytes, same cache policy), measured in that process
            "traffic_skeleton_own_policy_GBs": roof.get("traffic_skeleton_own_policy_GBs"),
            "frac_of_own_traffic_skeleton": roof.get("frac_of_own_traffic_skeleton"),
            "measured_in": "child process: " + " ".join(cmd[1:])}


def measured_ceilings(tsa, torch, dev, np_, nlev, elems):
    """What the memory system of THIS box delivers, next to the 8 TB/s spec peak (BASELINE.md
    section 3): the tuned device copy (16 B/lane, several loads in flight, best variant on this
    box; the guide's float4 copy reaches 6.29 TB/s), the naive 8 B/lane grid-stride copy that
    calibrates the PMC counters, and the traffic skeleton — exactly the bytes and addressing of
    the CAAR kernel with no arithmetic, non-temporal accesses."""
    import ctypes as C
    L = tsa.library()
    st = torch.cuda.current_stream(dev)

    def timed(fn, reps=10):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(reps):
            fn()
        e1.record(st)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / reps * 1e-3

    n = 1 << 27  # 1 GiB in + 1 GiB out per copy: far beyond the 256 MiB Infinity Cache
    src = torch.ones(n, dtype=torch.float64, device=dev)
    dst = torch.empty_like(src)
    sp, dp, sv = C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_void_p(st.cuda_stream)
    t = timed(lambda: L.check(L.lib.caar_stream_copy(dp, sp, n, 8, sv), "copy"))
    out = {"stream_copy_naive_GBs": 2 * n * 8 / t / 1e9}
    best = (0.0, -1)
    per_variant = []
    for v in range(L.lib.caar_stream_copy_tuned_variants()):
        t = timed(lambda: L.check(L.lib.caar_stream_copy_tuned(dp, sp, n, v, sv), "tuned copy"))
        gbs = 2 * n * 8 / t / 1e9
        per_variant.append(round(gbs, 1))
        best = max(best, (gbs, v))
    out["stream_copy_GBs"] = best[0]
    out["stream_copy_variant"] = L.lib.caar_stream_copy_tuned_info(best[1]).decode()
    out["stream_copy_all_variants_GBs"] = per_variant
    del src, dst
    if np_ == 4:
        data = tsa.TestData().init_data(elems, np_, nlev, device=dev)
        dims, ptrs, prm = data.arrays.dims(), data.arrays.pointers(), data.params()
        t = timed(lambda: L.check(L.lib.caar_traffic_skeleton(C.byref(dims), C.byref(ptrs), C.byref(prm), 22 if nlev == 72 else 2,
                                                              C.c_void_p(st.cuda_stream)), "skeleton"))
        out["traffic_skeleton_GBs"] = tsa.algorithmic_bytes(np_, nlev) * elems / t / 1e9
        if nlev == 72:
            # ... and with the HYBRID cache policy and window of the default kernel (replayed like the kernel: the kept
            # accumulators are found in the Infinity Cache): the arithmetic-free ceiling of `roofline.achieved` itself
            best = 0.0
            for v in (28, 29):
                for _ in range(3):
                    L.check(L.lib.caar_traffic_skeleton(C.byref(dims), C.byref(ptrs), C.byref(prm), v, C.c_void_p(st.cuda_stream)), "skeleton")
                t = timed(lambda: L.check(L.lib.caar_traffic_skeleton(C.byref(dims), C.byref(ptrs), C.byref(prm), v,
                                                                      C.c_void_p(st.cuda_stream)), "skeleton"), reps=20)
                best = max(best, tsa.algorithmic_bytes(np_, nlev) * elems / t / 1e9)
            out["traffic_skeleton_hybrid_GBs"] = best
        del data
    torch.cuda.empty_cache()
    return out


def own_traffic_skeleton(tsa, torch, data, dev, np_, nlev, elems):
    """The configuration's kernel against ITS OWN traffic with the arithmetic taken out (caar_traffic_skeleton: the same
    bytes, addressing and cache policy — NP=4: the hybrid policy with the current window, replayed like the kernel, variants
    28 / 29; NP=8: all-streaming, the kernel's 8 x 9 shape, variants 0 / 2) on the arrays of this process: GB/s of the best
    variant, or None where there is no skeleton for the configuration."""
    import ctypes as C
    L = tsa.library()
    st = torch.cuda.current_stream(dev)
    variants = (28, 29) if (np_ == 4 and nlev in (72, 128)) else ((0, 2) if (np_ == 8 and nlev == 72) else ())
    dims, ptrs, prm = data.arrays.dims(), data.arrays.pointers(), data.params()
    best = None
    for v in variants:
        call = lambda: L.check(L.lib.caar_traffic_skeleton(C.byref(dims), C.byref(ptrs), C.byref(prm), v, C.c_void_p(st.cuda_stream)), "skeleton")  # noqa: E731
        for _ in range(4):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20):
            call()
        e1.record(st)
        torch.cuda.synchronize(dev)
        gbs = tsa.algorithmic_bytes(np_, nlev) * elems / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e9
        best = gbs if best is None else max(best, gbs)
    return best


def static_traffic(np_, nlev, elems, with_source=False):
    """HBM bytes per launch from the committed PMC profile (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    separate passes, corrected as MI355X_MICROARCH.md prescribes; profiles/hbm_traffic.json, built by
    tools/make_hbm_traffic.py from the newest round's passes).  NOT measured in this run: counters
    need the profiler around the process."""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
        row = j.get("np%d_nlev%d_e%d" % (np_, nlev, elems), {})
        if with_source:
            return row.get("hbm_bytes_per_launch"), row.get("source"), row.get("kernel")
        return row.get("hbm_bytes_per_launch")
    except Exception:
        return (None, None, None) if with_source else None


TRAFFIC_SOURCE = ("static: profiles/hbm_traffic.json (rocprofv3 --pmc passes committed with the repo; "
                  "not measured in this run; L2-side counters: Infinity-Cache hits are counted as traffic)")
TRAFFIC_SOURCE_LIVE = ("measured in this run: two child passes of tools/pmc_run.py under rocprofv3 --pmc FETCH_SIZE / "
                       "--pmc WRITE_SIZE (separate passes, KiB units, each calibrated on the 8 B/lane stream copy of the same "
                       "pass whose byte count is known, as MI355X_MICROARCH.md prescribes; tools/pmc_parse.py); L2-side "
                       "counters: Infinity-Cache hits are counted as traffic")


STEP_CALLS = 20   # calls per launch of the driver-loop leg (roofline.run_steps) and of its counter pass


def live_traffic(np_, nlev, elems, timeout=240, steps=STEP_CALLS):
    """HBM-side bytes per launch of the default kernel, measured NOW: the counters need the profiler around the process,
    so two short child processes run the same launch (same data set size, fresh arrays) under rocprofv3 — one per counter,
    they do not fit one pass — and tools/pmc_parse.py turns them into bytes.  None if rocprofv3 is not there, this process
    is itself being profiled, or anything fails (the line then carries the committed figure and says so)."""
    import shutil
    import subprocess
    import tempfile
    if any("ROCPROF" in k or k.startswith("ROCP_") for k in os.environ) or shutil.which("rocprofv3") is None:
        return None
    tmp = tempfile.mkdtemp(prefix="caar_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            # python3 itself after `--`: no env / shell hop between the profiler and the program
            cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, ctr), "--",
                   sys.executable, os.path.join(ROOT, "tools", "pmc_run.py"), "--np", str(np_), "--nlev", str(nlev),
                   "--elems", str(elems), "--steps", str(steps)]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd="/tmp")
            if r.returncode != 0:
                return None
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_parse.py"), os.path.join(tmp, "FETCH_SIZE"),
                            os.path.join(tmp, "WRITE_SIZE")], capture_output=True, text=True, timeout=60)
        j = json.loads(r.stdout)
        return {"bytes": j["hbm_bytes_per_launch"], "read": j["caar_read_bytes_per_launch"],
                "write": j["caar_write_bytes_per_launch"],
                "calibration": {"FETCH_SIZE_x": j["counters"]["FETCH_SIZE"]["factor_8B_lane"],
                                "WRITE_SIZE_x": j["counters"]["WRITE_SIZE"]["factor_8B_lane"]},
                "kernel": j["counters"]["FETCH_SIZE"]["caar_kernel"],
                "steps": ({"bytes_per_launch": j["steps_hbm_bytes_per_launch"], "read_per_launch": j["steps_read_bytes_per_launch"],
                           "write_per_launch": j["steps_write_bytes_per_launch"], "calls_per_launch": steps,
                           "kernel": j["counters"]["FETCH_SIZE"]["steps_kernel"]}
                          if steps and "steps_hbm_bytes_per_launch" in j else None)}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def spin_up(tsa, torch, data, stream, dev, block=20, max_blocks=40, tol=0.003):
    """Untimed launches of the same step until its time per launch is stable.  A fresh process does not run at its
    steady rate at once: the first ~60 launches (~25 ms) of a 10 000-element step ramp from ~70 % to the steady 74-77 %
    of peak on this pool (power state, TLBs, the cache window filling; profiles/r02/cold_probe.log).  A time-stepping
    host runs in the steady state; W = 5 warmup steps (2 ms) do not reach it.  Returns the per-launch times of the
    blocks (ms): [0] is the cold figure."""
    blocks = []
    for _ in range(max_blocks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(block):
            tsa.compute_and_apply_rhs(data, stream)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        blocks.append(e0.elapsed_time(e1) / block)
        if len(blocks) >= 3 and abs(blocks[-1] - blocks[-2]) <= tol * blocks[-1] and abs(blocks[-2] - blocks[-3]) <= tol * blocks[-2]:
            break
    return blocks


def adaptive_state(lib, data):
    import ctypes as C
    if not lib.caar_get_adaptive_window():
        return {"enabled": False}
    w, st, n = C.c_double(0), C.c_double(0), C.c_longlong(0)
    state = lib.caar_adaptive_window_state(C.c_void_p(data.arrays["elem_derived_vn0"].data_ptr()), C.byref(w), C.byref(st), C.byref(n))
    return {"enabled": True, "policy_in_force": {1: "window", 0: "all_streaming", -1: "window (no hybrid kernel / not probed)"}[state],
            "probe_ms_window": w.value, "probe_ms_all_streaming": st.value, "probes": n.value}


def balg_of(tsa, args):
    return tsa.algorithmic_bytes(args.np_, args.nlev)


def time_launches(tsa, torch, data, stream, dev, steps, warmup):
    for _ in range(warmup):
        tsa.compute_and_apply_rhs(data, stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(steps):
        tsa.compute_and_apply_rhs(data, stream)
    e1.record(stream)
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) / steps


def placement_spread(tsa, torch, args, data, dev, stream, mine, nets):
    """The steady rate depends on WHERE the driver placed the arrays in HBM: allocations of the same process differ by a
    reproducible 3-5 % (two levels, e.g. 73.5 / 76.5 % of peak; not TLB misses, not the cache window, not the relative
    offsets of the arrays: DESIGN.md section 5, profiles/r02/placement_probes.log), and allocations made after a lot of
    allocate/free traffic tend to get the slower level.  The library's allocator (caar_arrays_alloc, used by
    TestData.init_data) spreads every array over the address classes by construction.  `value` is measured on the first
    allocation of the process; this repeats the same step on it, on two more allocations of that kind and on one set of
    plain torch allocations (last entry), so the line shows what placement is worth in this process."""
    out = []
    others = [tsa.TestData().init_data(mine, args.np_, args.nlev, device=dev, first_elem=nets) for _ in range(2)]
    # ... and on torch's own sixteen allocations (what round 1 measured on), last in the list
    others.append(tsa.TestData().init_data(mine, args.np_, args.nlev, device=dev, first_elem=nets, place="torch"))
    for d in [data] + others:
        time_launches(tsa, torch, d, stream, dev, 20, 40)
        # best of three blocks: a block is occasionally hit by a stall of tens of ms (the driver wiping memory freed
        # earlier in the run), which says nothing about the placement
        ms = min(time_launches(tsa, torch, d, stream, dev, 20, 0) for _ in range(3))
        out.append(balg_of(tsa, args) * mine / (ms * 1e-3) / 1e9)
    del others
    torch.cuda.empty_cache()
    return out


def interleaved_sequences(tsa, torch, args, data, dev, stream, mine, steps):
    """What a time-stepping HOST gets, as opposed to bench.py's replay loop: compute_and_apply_rhs alternated with other
    work on the same elements, with TestData::update_time_levels between the steps (the reference's driver loop with its
    commented-out rotation, main.cpp:113-121, :118).  Two neighbours are measured:
      * "euler_step": caar_euler_step with 4 tracers on the same elements (EulerStepFunctor.hpp:32-68; this library's own
        kernel, all accesses non-temporal: 0.92 GB of traffic per call at 10 000 elements, none of it allocating in the
        Infinity Cache);
      * "evicting": a foreign kernel with the default cache policy that moves 768 MiB (torch.add on 256 MiB operands) —
        three times the Infinity Cache, so whatever the CAAR kernel left there is gone when it runs again.
    The CAAR time inside a sequence is (time of K steps of the sequence - time of K calls of the neighbour alone) / K: all
    interference is charged to CAAR.  Every sequence is run with the default kernel and with its all-streaming twin
    (variant 1), so the line shows what the cache window is worth under each host pattern.  Outside the timed region of
    `value`."""
    import ctypes as C
    L = tsa.library()
    lib = L.lib
    np_, nlev = args.np_, args.nlev
    f64 = torch.float64
    qsize = 4
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    vstar = torch.rand((mine, nlev, np_, np_, 2), dtype=f64, device=dev, generator=g)
    qdp4 = torch.rand((mine, qsize, 2, nlev, np_, np_), dtype=f64, device=dev, generator=g)
    qtens = torch.empty((mine, qsize, nlev, np_, np_), dtype=f64, device=dev)
    geo = {n: data.arrays["elem_" + n] for n in ("Dinv", "metdet", "rmetdet")}
    dvv = data.dvv_device()
    n_ev = 1 << 25  # 256 MiB per operand
    ev_a = torch.ones(n_ev, dtype=f64, device=dev)
    ev_b = torch.ones(n_ev, dtype=f64, device=dev)
    ev_c = torch.empty(n_ev, dtype=f64, device=dev)
    neighbours = {
        "euler_step": (lambda: tsa.euler_step(vstar, qdp4, geo, dvv, qsize, 0, 0.01, data.constants.rrearth, out=qtens),
                       "caar_euler_step, 4 tracers, same elements (nt accesses, %.2f GB per call)" % (
                           (vstar.numel() + qdp4.numel() // 2 + qtens.numel()) * 8 / 1e9)),
        "evicting": (lambda: torch.add(ev_a, ev_b, out=ev_c),
                     "torch.add on three 256 MiB arrays (default cache policy, 0.81 GB per call: evicts the Infinity Cache)"),
    }
    saved = (data.control.n0, data.control.np1, data.control.nm1)

    def timed(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n):
            fn()
        e1.record(stream)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / n

    def step_with(other):
        def f():
            tsa.compute_and_apply_rhs(data, stream)
            other()
            data.update_time_levels()
        return f

    balg_launch = tsa.algorithmic_bytes(np_, nlev) * mine
    out = {}
    have_twin = np_ == 4 and lib.caar_num_variants(np_, nlev) > 1
    try:
        for name, (other, what) in neighbours.items():
            timed(other, 5)
            other_ms = min(timed(other, steps) for _ in range(2))
            row = {"neighbour": what, "neighbour_ms": other_ms, "steps": steps}
            for label, variant in (("default", 0), ("all_streaming", 1)):
                if variant and not have_twin:
                    continue
                lib.caar_select_variant(np_, nlev, variant)
                lib.caar_adaptive_window_reset()   # this leg is a new host pattern: the library starts from its default
                seq = step_with(other)
                # the default kernel's adaptive window (include/caar.h) needs its first probe (48 calls) or a drift + probe
                # (~45 calls) to settle on the policy this pattern wants: the sequence runs untimed until then
                adapt = 160 if (variant == 0 and lib.caar_get_adaptive_window()) else 10
                timed(seq, adapt)
                seq_ms = min(timed(seq, steps) for _ in range(2))
                caar_ms = seq_ms - other_ms
                row[label] = {"sequence_ms": seq_ms, "caar_ms": caar_ms, "achieved": balg_launch / (caar_ms * 1e-3) / 1e9,
                              "frac": balg_launch / (caar_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "kernel": lib.caar_kernel_name(np_, nlev).decode(), "untimed_calls_before": adapt}
                if variant == 0:
                    state = lib.caar_adaptive_window_state(C.c_void_p(data.arrays["elem_derived_vn0"].data_ptr()), None, None, None)
                    row[label]["policy_in_force"] = {1: "window", 0: "all_streaming (adapted)", -1: "window (not adaptive)"}[state]
            out[name] = row
    finally:
        lib.caar_select_variant(np_, nlev, 0)
        lib.caar_adaptive_window_reset()
        data.control.n0, data.control.np1, data.control.nm1 = saved
    del vstar, qdp4, qtens, ev_a, ev_b, ev_c
    torch.cuda.empty_cache()
    return out


def subrange_streams_leg(tsa, torch, args, data, dev, stream, mine, nstreams=4, steps=20):
    """The reference's horizontal-OpenMP pattern on one GPU (data_structures.hpp:58-69: every host thread calls the routine
    with its own Control{nets, nete}): `nstreams` streams, each stepping its own contiguous quarter of the SAME arrays, side
    by side.  The hybrid policy's cache window is a budget of the device (element_is_cached keys on the element's index in
    the arrays), so the four launches together keep one window's worth; round 3 budgeted per launch — emulated here by a
    window `nstreams` times as large — and all-streaming is the floor.  Time = `steps` rounds of all sub-ranges, main-stream
    events around the fork/join.  Outside the timed region of `value`."""
    import copy
    lib = tsa.library().lib
    np_, nlev = args.np_, args.nlev
    side = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    parts = []
    for i in range(nstreams):
        d = copy.copy(data)
        d.control = copy.copy(data.control)
        d.control.nets, d.control.nete = tsa.shard_range(mine, i, nstreams)
        parts.append(d)
    data.dvv_device()

    def rounds(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for s_ in side:
            s_.wait_stream(stream)
        for _ in range(n):
            for d, s_ in zip(parts, side):
                tsa.compute_and_apply_rhs(d, s_)
        for s_ in side:
            stream.wait_stream(s_)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / n

    balg_round = tsa.algorithmic_bytes(np_, nlev) * mine
    window = int(lib.caar_get_cache_window())
    out = {"streams": nstreams, "elements_per_stream": [p_.control.nete - p_.control.nets for p_ in parts], "steps": steps}
    have_twin = np_ == 4 and lib.caar_num_variants(np_, nlev) > 1
    try:
        for label, variant, win in (("device_budget", 0, window), ("per_launch_budget_r03", 0, window * nstreams),
                                    ("all_streaming", 1, window)):
            if variant and not have_twin:
                continue
            lib.caar_select_variant(np_, nlev, variant)
            lib.caar_set_cache_window(win)
            rounds(10)
            ms = min(rounds(steps) for _ in range(3))
            out[label] = {"ms_per_round": ms, "achieved": balg_round / (ms * 1e-3) / 1e9,
                          "frac": balg_round / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "cache_window_bytes": win if not variant else 0}
    finally:
        lib.caar_select_variant(np_, nlev, 0)
        lib.caar_set_cache_window(window)
    return out


FP64_VECTOR_PEAK_TFLOPS = 78.6   # 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz = half the guide's FP32 vector peak (157.3); SURVEY 8d


def _kernel_key(name):
    """'void caar::k<1, 2>(args) [clone]' -> 'k<1,2>' (rocprofv3 prints the full signature, the profiles keep the bare name)"""
    name = name.strip()
    if name.startswith("void "):
        name = name[5:]
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return name[:cut].replace("caar::", "").replace(" ", "")


def issue_profile(kernel_name):
    """The committed instruction-issue counters of the step-loop kernels (profiles/*/pmc_issue.json, tools/pmc_issue.py:
    rocprofv3 --pmc SQ_* passes): static, NOT measured in this run.  Newest round first; None if the kernel is not there."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_issue.json")), reverse=True):
        try:
            j = json.load(open(path))
        except Exception:
            continue
        for k, row in j.items():
            if _kernel_key(k) == _kernel_key(kernel_name) and "valu_issue_utilisation" in row:
                out = {n: row[n] for n in ("valu_issue_utilisation", "mfma_pipe_utilisation", "wave_cycles_waiting_for_any_instruction",
                                           "lds_bank_conflict_over_lds_active_cycles", "waves_per_simd") if n in row}
                out["valu_instructions_per_element_call"] = row.get("SQ_INSTS_VALU", {}).get("per_element_call")
                out["source"] = "static: %s (committed rocprofv3 --pmc SQ_* passes, tools/pmc_issue.py; not measured in this run)" % \
                    os.path.relpath(path, ROOT)
                return out
    return None


def run_steps_leg(tsa, torch, args, data, dev, stream, mine, calls=STEP_CALLS):
    """The driver loop itself (main.cpp:113-121 with update_time_levels): `calls` calls with rotating time levels through
    caar_launch_steps — ONE launch where a step-loop kernel exists (every workgroup makes all the calls for its element, the
    prognostic state carried from call to call in registers / LDS; bit-identical to single launches) — against
    the same calls launched one by one.  Outside the timed region of `value`; the headline is per call and does not use it.

    Its own roofline: the loop does NOT move the algorithmic bytes of `calls` single calls (state that stays on chip is
    neither re-read nor stored, dead stores are dropped), so dividing those bytes by its time gives a figure above the HBM
    peak that is no bandwidth.  What is reported instead: the time against the HBM bound of single calls (a time), the
    fp64 rate against the vector peak (the loop is bound by instruction issue at two waves per SIMD), and — filled in by
    main() from the live counter pass — the HBM bytes the step-loop kernel really moves per call."""
    lib = tsa.library().lib
    saved = (data.control.n0, data.control.np1, data.control.nm1, data.control.dt2, data.constants.eta_ave_w)
    data.control.dt2, data.constants.eta_ave_w = 1.0e-6, 0.0   # timing only: keeps hundreds of leap-frog steps finite

    def timed(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            tsa.compute_and_apply_rhs_steps(data, calls, True, stream)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / (reps * calls)

    out = {"calls_per_launch": calls, "one_launch_available": bool(lib.caar_has_fused_steps(args.np_, args.nlev, 0))}
    try:
        for label, fused in (("ms_per_call_single_launches", 0), ("ms_per_call_one_launch", 1)):
            lib.caar_set_fused_steps(fused)
            timed(2)
            out[label] = min(timed(3) for _ in range(2))
    finally:
        lib.caar_set_fused_steps(1)
        (data.control.n0, data.control.np1, data.control.nm1, data.control.dt2, data.constants.eta_ave_w) = saved
    ms = out["ms_per_call_one_launch"]
    out["speedup"] = out["ms_per_call_single_launches"] / ms
    out["element_updates_per_s_one_launch"] = mine / (ms * 1e-3)
    flops = args.nlev * (20 * args.np_ ** 3 + 124 * args.np_ ** 2)        # algorithmic flops per element-update
    tf = flops * mine / (ms * 1e-3) / 1e12
    out["roofline"] = {
        "bound": "valu_issue",
        "achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_VECTOR_PEAK_TFLOPS,
        "flops_per_element_call": flops,
        "ms_per_call_hbm_bound_of_single_calls": tsa.algorithmic_bytes(args.np_, args.nlev) * mine / (HBM_PEAK_GBS * 1e9) * 1e3,
        "hbm_bytes_per_call": None, "hbm_GBs": None, "hbm_frac_of_peak": None,   # main(): live counter pass of the step-loop kernel
        "issue": None,   # main(): the committed SQ-counter profile of the kernel the live pass saw
        "note": "one launch makes all %d calls per element with the prognostic state on chip: its HBM traffic per call is a fraction "
                "of a single call's algorithmic bytes (hbm_bytes_per_call, measured), so the single-call HBM bound "
                "(ms_per_call_hbm_bound_of_single_calls) does not apply to it; results bit-identical to single launches" % calls,
    }
    return out


def dropin_leg(np_, nlev, elems, kernel_ms, timeout=300):
    """The literal drop-in: the reference driver's loop (main.cpp:113-121) calling Homme::compute_and_apply_rhs(TestData&)
    on arrays the HOST owns, through this repo's shim (tinman_sandbox_amd/host/homme_caar.cpp, the file the reference's
    unchanged main.cpp links against: oracle/Makefile pointers_only_hip) — run by this repo's own driver binary
    (host/caar_driver --tinman-host-arrays=yes: the same shim, the same loop) in two child processes:
      mapped    the default: the kernel works on the page-locked host arrays over PCIe, synchronous per call;
      resident  CAAR_SHIM_RESIDENT=1: uploaded at the first call, later calls only enqueue the kernel.
    ms per call excludes the first call's page-locking / upload (Homme::shim_stats).  Outside `value`."""
    import re
    import subprocess
    suffix = "" if (np_, nlev) == (4, 72) else "_np%d_nlev%d" % (np_, nlev)
    exe = os.path.join(ROOT, "tinman_sandbox_amd", "host", "caar_driver" + suffix)
    if not os.path.exists(exe):
        return {"error": "host/caar_driver%s is not built" % suffix}
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "CAAR_SHIM_RESIDENT"):
        env.pop(k, None)
    out = {"driver": "tinman_sandbox_amd/host/caar_driver%s --tinman-host-arrays=yes (Homme::compute_and_apply_rhs(TestData&) in the "
                     "reference's loop, main.cpp:113-121)" % suffix, "elements": elems}
    norms = {}
    for mode, calls, extra in (("mapped", 12, []), ("resident", 400, ["--tinman-resident=yes"])):
        cmd = [exe, "--tinman-num-elems=%d" % elems, "--tinman-num-exec=%d" % calls, "--tinman-host-arrays=yes"] + extra
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        except subprocess.TimeoutExpired:
            out[mode] = {"error": "timeout"}
            continue
        m = re.search(r"shim_stats mode=(\w+) calls=(\d+) seconds=([-+0-9.eE]+) wall=([-+0-9.eE]+)", r.stdout)
        if r.returncode != 0 or not m or m.group(1) != mode:
            out[mode] = {"error": (r.stderr or r.stdout)[-300:]}
            continue
        n, secs = int(m.group(2)), float(m.group(3))
        ms = 1e3 * secs / n
        out[mode] = {"calls": n, "ms_per_call": ms, "element_updates_per_s": elems / (ms * 1e-3),
                     "over_kernel_ms": ms / kernel_ms if kernel_ms else None,
                     "loop_wall_seconds_incl_first_call": float(m.group(4))}
        norms[mode] = re.findall(r"\|\|(?:v|T|dp)\|\|_2\s*=\s*([-+0-9.eE]+)", r.stdout)[-3:]
    if len(norms) == 2:
        out["final_norms_equal_digit_for_digit"] = norms["mapped"] == norms["resident"]
    out["note"] = ("mapped is PCIe-bound by construction (every input byte crosses the link once per call); resident runs the "
                   "kernel of `value` behind the reference's signature; the first ~60 launches of the fresh child process ramp up "
                   "(roofline.spinup), which its %d calls include" % 400)
    return out


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process starts the N ranks itself and only
    waits for them.  It imports neither torch nor the library, so nothing here has initialised the GPU (no exec of an
    initialised process, no fork of one: the ranks are fresh interpreters).  Rank 0 inherits stdout — its ONE JSON line
    is this command's output —, the other ranks' stdout goes to stderr.  A rank that fails takes the job down: the
    others (which would wait at a barrier) are terminated by their own PIDs after a short grace, and the exit code is
    the first non-zero one."""
    import signal
    import socket
    import subprocess
    n = args.gpus
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CAAR_BENCH_LAUNCHER="bench.py")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc, failed_at = 0, None
    try:
        live = list(procs)
        while live:
            for p in list(live):
                c = p.poll()
                if c is None:
                    continue
                live.remove(p)
                if c != 0 and rc == 0:
                    rc, failed_at = c, time.monotonic()
            if failed_at is not None and live and time.monotonic() - failed_at > 15.0:
                for p in live:
                    p.terminate()
                for p in live:
                    try:
                        p.wait(10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                break
            time.sleep(0.05)
    except KeyboardInterrupt:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGINT)
        rc = 130
    for p in procs:
        p.wait()
    return rc if rc >= 0 else 1


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import torch
    import tinman_sandbox_amd as tsa

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs MI355X GPUs (no CPU fallback)")
    # one rank per GPU; CAAR_BENCH_BACKEND=gloo is a rehearsal mode that lets several ranks
    # share the GPUs that exist (used to exercise the N>1 path on a one-GPU box)
    backend = os.environ.get("CAAR_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= ndev:
        sys.exit("rank %d has no GPU (%d visible)" % (local_rank, ndev))
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    dist = None
    # CAAR_BENCH_FORCE_DIST=1: a process group even for one rank (under torch.distributed.run --nproc-per-node 1), so that
    # the RCCL path of the N>1 line — init with device_id, barriers, the MAX all-reduce and the gather on device tensors —
    # can be exercised on a one-GPU box (tests/test_bench_gpu.py)
    if world > 1 or os.environ.get("CAAR_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:   # a one-rank group started plainly (CAAR_BENCH_FORCE_DIST=1 python bench.py)
            import socket
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
            s.close()
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    reduce_dev = dev if backend == "nccl" else torch.device("cpu")
    lib = tsa.library().lib

    # ---- this rank's slab of the global element range -------------------------------
    if args.total_elems is not None:
        total_elems = args.total_elems
    else:
        per_gpu = args.elems_per_gpu if args.elems_per_gpu is not None else (10000 if world == 1 else 12500)
        total_elems = per_gpu * world
    nets, nete = tsa.shard_range(total_elems, rank, world)
    mine = nete - nets
    data = tsa.TestData().init_data(mine, args.np_, args.nlev, device=dev, first_elem=nets)
    stream = torch.cuda.current_stream(dev)

    def barrier():
        if dist is not None:
            dist.barrier()

    # first-use cost (code object load, first-touch) and, unless --no-spinup, the ramp to the steady state
    tsa.compute_and_apply_rhs(data, stream)
    torch.cuda.synchronize(dev)
    spin = [] if args.no_spinup else spin_up(tsa, torch, data, stream, dev)

    for _ in range(args.warmup):
        tsa.compute_and_apply_rhs(data, stream)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        tsa.compute_and_apply_rhs(data, stream)
    ev1.record(stream)
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0          # this rank's K steps, start barrier to its own last step done
    barrier()
    torch.cuda.synchronize(dev)
    wall_with_barrier = time.perf_counter() - t0  # ... plus the closing barrier (collective latency, not step time)
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream

    from tinman_sandbox_amd import sharding
    # the job is done when its slowest rank is: MAX over ranks of each rank's own time for the K steps.  All ranks leave
    # the opening barrier together; the closing barrier's own latency (tens of microseconds to milliseconds over 8 ranks,
    # against ~8 ms of timed steps) is reported beside it, not charged to the steps.
    wall_max, kernel_ms_max, wall_with_barrier_max = sharding.max_over_ranks([wall, kernel_ms, wall_with_barrier], dist,
                                                                             reduce_dev)
    per_rank = sharding.gather_over_ranks([float(mine), kernel_ms], dist, reduce_dev)  # [[elems, ms], ...]

    # The default NP=4 kernels keep the read-modify-write accumulators of part of the elements in the
    # Infinity Cache between calls (caar_set_cache_window): bench.py replays the same arrays back to back,
    # so those bytes are served on-chip and `achieved` above is ALGORITHMIC throughput, not DRAM
    # throughput.  The same launches with every access streaming (window 0), outside the timed region:
    kernel_ms_streaming = None
    if args.np_ == 4 and lib.caar_num_variants(args.np_, args.nlev) > 1:
        # variant 1 of every NP=4 table is the all-streaming (non-temporal, no cache window) form of variant 0:
        # same launch shape, another kernel name, so a rocprofv3 --stats run of this command keeps the two apart
        lib.caar_select_variant(args.np_, args.nlev, 1)
        streaming_kernel = lib.caar_kernel_name(args.np_, args.nlev).decode()
        kernel_ms_streaming = time_launches(tsa, torch, data, stream, dev, args.steps, 3)
        lib.caar_select_variant(args.np_, args.nlev, 0)
        (kernel_ms_streaming,) = sharding.max_over_ranks([kernel_ms_streaming], dist, reduce_dev)

    # per-launch spread ("median and min"), outside the timed region, two ways:
    #  * back to back: the K steps once more with an event recorded after every launch; interval i is launch i's time under
    #    the conditions of the timed region (its predecessor's write-back still draining) -> kernel_ms_median / kernel_ms_min;
    #  * isolated: one event pair per launch, the stream idle before each (~4 % shorter: a launch that starts on an idle
    #    chip does not share HBM with its predecessor's dirty lines) -> kernel_ms_isolated_*.
    # `value` and roofline.achieved use the mean of the timed region above.  --no-isolated skips both (a rocprofv3 --stats
    # run then sees only spin-up, warm-up and timed launches of this grid; tools/trace_stats.py --window cuts either way).
    per_launch, back_to_back = [], []
    if rank == 0 and not args.no_isolated:
        n = max(args.steps, 20)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        tsa.compute_and_apply_rhs(data, stream)
        evs[0].record(stream)
        for i in range(n):
            tsa.compute_and_apply_rhs(data, stream)
            evs[i + 1].record(stream)
        torch.cuda.synchronize(dev)
        back_to_back = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n))
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in pairs:
            a.record(stream)
            tsa.compute_and_apply_rhs(data, stream)
            b.record(stream)
            torch.cuda.synchronize(dev)
        per_launch = sorted(a.elapsed_time(b) for a, b in pairs)

    if rank == 0:
        balg = tsa.algorithmic_bytes(args.np_, args.nlev)
        per_launch_bytes = balg * mine
        achieved = per_launch_bytes / (kernel_ms_max * 1e-3) / 1e9
        window = int(lib.caar_get_cache_window())
        roof = {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": static_traffic(args.np_, args.nlev, mine),
            "traffic_source": TRAFFIC_SOURCE + ("; this configuration's passes: %s, kernel %s" % static_traffic(
                args.np_, args.nlev, mine, True)[1:] if static_traffic(args.np_, args.nlev, mine) else ""),
            "algorithmic_bytes_per_element": balg,
            "elements_per_launch": mine,
            "kernel_ms": kernel_ms_max,
            "cache_window_bytes": window if args.np_ == 4 else 0,
            # the adaptive window (include/caar.h): the policy the library settled on for these arrays during spin-up and
            # the kernel times of its probe (ms; 7 calls of each policy, medians of the last 4)
            "cache_window_adaptive": adaptive_state(lib, data),
            "achieved_note": "algorithmic bytes / kernel time with the default hybrid cache policy: the "
                             "accumulators of part of the elements stay in the Infinity Cache between the "
                             "back-to-back calls, so DRAM traffic is lower than the algorithmic bytes; "
                             "achieved_all_streaming is the all-streaming variant of the same launch shape "
                             "(caar_select_variant 1: every access non-temporal, no cache window)"
                             if args.np_ == 4 and window else "every access streams from/to HBM",
            "kernel_ms_median": back_to_back[len(back_to_back) // 2] if back_to_back else None,
            "kernel_ms_min": back_to_back[0] if back_to_back else None,
            "kernel_ms_max": back_to_back[-1] if back_to_back else None,
            "frac_median": per_launch_bytes / (back_to_back[len(back_to_back) // 2] * 1e-3) / 1e9 / HBM_PEAK_GBS if back_to_back else None,
            "kernel_ms_isolated_min": per_launch[0] if per_launch else None,
            "kernel_ms_isolated_median": per_launch[len(per_launch) // 2] if per_launch else None,
            # the untimed spin-up before the W warmup steps (see spin_up): how long it took to reach the steady
            # rate, and what the first block of launches of this process ran at (the cold figure)
            # which dispatches of this kernel on this grid are the timed ones (0-based, in launch order), so that a
            # rocprofv3 kernel trace of this command can be cut to the timed region (tools/trace_stats.py --window)
            "timed_dispatches": [1 + 20 * len(spin) + args.warmup, 1 + 20 * len(spin) + args.warmup + args.steps],
            "spinup": {"launches": 20 * len(spin), "kernel_ms_first_20_launches": spin[0],
                       "achieved_first_20_launches": per_launch_bytes / (spin[0] * 1e-3) / 1e9,
                       "kernel_ms_last_20_launches": spin[-1]} if spin else None,
        }
        if args.np_ == 4 and window:
            # How much of `achieved` can be DRAM traffic: the hybrid policy keeps vn0 (2 blocks), omega_p and eta_dot_dpdn of
            # `kept` elements in the Infinity Cache, and a kept byte saves one read and one write per call.  If every one of
            # them hits (the replay loop of this benchmark: nothing evicts them), DRAM sees the algorithmic bytes minus that.
            pp = args.np_ * args.np_
            keep_per_elem = 8 * (4 * pp * args.nlev + pp)
            kept = min(mine, window // keep_per_elem)
            dram = per_launch_bytes - 2 * kept * keep_per_elem
            roof["dram_estimate"] = {
                "elements_kept": int(kept), "bytes_kept_per_element": keep_per_elem,
                "dram_bytes_per_launch_if_all_kept_bytes_hit": dram,
                "achieved": dram / (kernel_ms_max * 1e-3) / 1e9, "frac": dram / (kernel_ms_max * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "lower bound of the DRAM rate behind `achieved` (an estimate: no counter on this pool separates Infinity-Cache "
                        "hits from DRAM; roofline.traffic is measured in front of the cache)"}
        if kernel_ms_streaming is not None:
            a0 = per_launch_bytes / (kernel_ms_streaming * 1e-3) / 1e9
            roof["achieved_all_streaming"] = a0
            roof["frac_all_streaming"] = a0 / HBM_PEAK_GBS
            roof["kernel_ms_all_streaming"] = kernel_ms_streaming
            roof["kernel_all_streaming"] = streaming_kernel
        if dist is not None:
            roof["per_gpu"] = [{"rank": r, "elements": int(e), "kernel_ms": ms,
                                "achieved": balg * e / (ms * 1e-3) / 1e9,
                                "frac": balg * e / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
                               for r, (e, ms) in enumerate(per_rank)]
        if world == 1:
            # the kernel against ITS OWN traffic with the arithmetic taken out, on the same arrays, same cache policy and window,
            # replayed like the kernel (outside the timed region; the skeleton's stores only touch what the next call overwrites)
            sk = own_traffic_skeleton(tsa, torch, data, dev, args.np_, args.nlev, mine)
            if sk:
                roof["traffic_skeleton_own_policy_GBs"] = sk
                roof["frac_of_own_traffic_skeleton"] = achieved / sk
        out = {
            "metric": "element-RHS-updates/sec (node) + achieved HBM GB/s, NP=%d NLEV=%d fp64" % (args.np_, args.nlev),
            "value": total_elems * args.steps / wall_max,
            "unit": "element-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            # what actually ran before the timed region: the first-use launch, the untimed spin-up to the steady state
            # (see spin_up; --no-spinup: none) and the W warmup steps
            "warmup_effective": 1 + 20 * len(spin) + args.warmup,
            "ms_per_step": wall_max / args.steps * 1e3,
            "ms_per_step_incl_closing_barrier": wall_with_barrier_max / args.steps * 1e3,
            # how the ranks were coupled: the torch.distributed backend ("nccl" is RCCL on ROCm; "gloo" only in the
            # one-GPU rehearsal mode CAAR_BENCH_BACKEND=gloo) and the world size the process group reports
            "backend": (dist.get_backend() if dist is not None else "none"),
            "dist_world_size": (dist.get_world_size() if dist is not None else 1),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "compute_and_apply_rhs NP=%d NLEV=%d num_elems=%d per GPU (%d total), moist, "
                            "reference closed-form element arrays" % (args.np_, args.nlev, mine, total_elems),
                "parallelism": "element-sharded x%d, no collectives" % world,
                "launcher": os.environ.get("CAAR_BENCH_LAUNCHER",
                                           "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else
                                           ("external" if "WORLD_SIZE" in os.environ else "none")),
                "kernel": lib.caar_kernel_name(args.np_, args.nlev).decode(),
                "array_placement": ("caar_arrays_alloc: 64 MiB physical chunks sampled from a temporary pool, every array spread "
                                    "over the device's address classes (DESIGN.md section 5)"
                                    if data.arrays.arena is not None and data.arrays.arena.spread() else "plain allocations"),
            },
            "hbm_gbs_algorithmic_job": total_elems * args.steps * balg / wall_max / 1e9,
            "roofline": roof,
        }
        if world > 1:
            # the N=1 line holds another per-GPU size (10 000 = configs[1]); the N=1 figure that is comparable with this
            # line — same kernel, same elements per GPU — is measured by the N=1 run in a child process:
            out["config"]["n1_reference"] = {
                "where": "other_configs[0] of `python bench.py --gpus 1` (element_updates_per_s, kernel_ms, frac_of_hbm_peak)",
                "workload": "NP=%d NLEV=%d num_elems=%d" % (args.np_, args.nlev, mine),
                "why": "--gpus 1 is BASELINE configs[1] (10 000 elements); --gpus N>1 holds %d per GPU so that --gpus 8 is "
                       "configs[2]; efficiency against the headline N=1 value mixes in the size effect" % mine,
            }
        if world == 1 and not args.no_other_configs and (args.np_, args.nlev) == (4, 72):
            # kernel-only, same run; the headline stays configs[1]:
            #  * 12 500 elements NP=4 NLEV=72 = one GPU's share of configs[2] (what --gpus 8 runs per GPU)
            #  * configs[3] NP=4 NLEV=128, one GPU's share of 100 000 elements
            #  * configs[4] NP=8, 20 000 elements
            # (the headline arrays stay allocated: freeing and re-allocating them would change what placement_spread sees)
            out["other_configs"] = [measure_config(4, 72, 12500, 20, 5), measure_config(4, 128, 12500, 20, 5),
                                    measure_config(8, 72, 20000, 10, 3)]
        if world == 1 and not args.no_live_traffic:
            # the arrays of this process stay allocated: the children hold their own (2-4 GB) for a few seconds
            lt = live_traffic(args.np_, args.nlev, mine)
            if lt is not None:
                roof["traffic"] = lt["bytes"]
                roof["traffic_source"] = TRAFFIC_SOURCE_LIVE
                roof["traffic_detail"] = lt
                roof["traffic_over_algorithmic"] = lt["bytes"] / per_launch_bytes
        if world == 1 and not args.no_steps_leg:
            # caar_run_steps / caar_launch_steps: not part of `value`
            roof["run_steps"] = rs = run_steps_leg(tsa, torch, args, data, dev, stream, mine)
            st = (roof.get("traffic_detail") or {}).get("steps")
            if st and rs["one_launch_available"]:
                per_call = st["bytes_per_launch"] / st["calls_per_launch"]
                gbs = per_call / (rs["ms_per_call_one_launch"] * 1e-3) / 1e9
                rs["roofline"].update({"hbm_bytes_per_call": per_call, "hbm_GBs": gbs, "hbm_frac_of_peak": gbs / HBM_PEAK_GBS,
                                       "hbm_bytes_per_element_call": per_call / mine,
                                       "hbm_bytes_over_single_call_algorithmic": per_call / per_launch_bytes,
                                       "hbm_kernel": st["kernel"], "hbm_source": TRAFFIC_SOURCE_LIVE,
                                       "issue": issue_profile(st["kernel"])})
        if world == 1 and not args.no_interleaved:
            seqs = interleaved_sequences(tsa, torch, args, data, dev, stream, mine, args.steps)
            roof["interleaved"] = seqs
            # CAAR inside the reference's driver loop with rotation and a tracer step between the calls ...
            roof["achieved_interleaved"] = seqs["euler_step"]["default"]["achieved"]
            roof["frac_interleaved"] = seqs["euler_step"]["default"]["frac"]
            # ... and with a neighbour that wipes the Infinity Cache between the calls
            roof["achieved_interleaved_evicting"] = seqs["evicting"]["default"]["achieved"]
            roof["frac_interleaved_evicting"] = seqs["evicting"]["default"]["frac"]
        if world == 1 and not args.no_interleaved and args.np_ == 4:
            # four host threads / streams on disjoint quarters of the arrays (the reference's horizontal-OpenMP pattern)
            roof["subrange_streams"] = subrange_streams_leg(tsa, torch, args, data, dev, stream, mine, steps=args.steps)
        if world == 1 and not args.no_other_configs:
            # the same step on [the timed allocation, two more placed allocations, plain torch allocations] (GB/s)
            roof["placement_spread_achieved"] = placement_spread(tsa, torch, args, data, dev, stream, mine, nets)
            del data
            torch.cuda.empty_cache()
            roof["measured_on_this_box"] = measured_ceilings(tsa, torch, dev, args.np_, args.nlev, mine)
            # against what THIS box's memory system gives a plain copy (best tuned variant), next to the 8 TB/s nominal peak
            copy = roof["measured_on_this_box"]["stream_copy_GBs"]
            roof["frac_of_measured_copy"] = roof["achieved"] / copy
            if "achieved_all_streaming" in roof:
                roof["frac_of_measured_copy_all_streaming"] = roof["achieved_all_streaming"] / copy
            # (measured_on_this_box.traffic_skeleton_hybrid_GBs is the hybrid skeleton on a FRESH allocation made at the end of
            # this process: placement differs from the timed arrays' by up to 5 %; the like-for-like ratio is
            # roofline.frac_of_own_traffic_skeleton, measured on the timed arrays themselves)
        if world == 1 and not args.no_dropin and not args.no_other_configs:
            out["dropin"] = dropin_leg(args.np_, args.nlev, mine, kernel_ms_max)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.np_, args.nlev, args.cpu_seconds, mine)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
