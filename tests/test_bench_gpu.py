"""bench.py's JSON line: the fields the driver and the judge read, on one GPU and in the two-rank rehearsal mode
(CAAR_BENCH_BACKEND=gloo: several ranks share the GPU that exists; the N>1 code path — slabs, barriers, max-over-ranks
timing, per-rank gather — is the one `--gpus 8` runs under RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]   # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    return env


def test_single_gpu_line_carries_the_contract_and_the_host_sequences():
    r = subprocess.run([sys.executable, BENCH, "--elems-per-gpu", "1500", "--steps", "4", "--warmup", "2", "--no-spinup",
                        "--no-other-configs", "--cpu-seconds", "0.2"], capture_output=True, text=True, timeout=900,
                       env=_clean_env())
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = _line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 4 and j["warmup"] == 2 and j["dtype"] == "f64" and j["vs_baseline"] is None
    assert j["backend"] == "none" and j["dist_world_size"] == 1
    assert j["warmup_effective"] == 1 + 2                      # first-use launch + W (no spin-up asked for)
    assert j["ms_per_step"] <= j["ms_per_step_incl_closing_barrier"]
    assert abs(j["value"] - 1500 / (j["ms_per_step"] * 1e-3)) <= 1e-6 * j["value"]
    roof = j["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and roof["unit"] == "GB/s"
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert roof["kernel_ms"] <= j["ms_per_step"] * 1.0001      # the kernel cannot take longer than the step that holds it
    # the three figures: replay, all-streaming, inside a host sequence (tracer step / cache-evicting neighbour)
    for k in ("achieved", "achieved_all_streaming", "achieved_interleaved", "achieved_interleaved_evicting"):
        assert roof[k] > 0, k
    for name in ("euler_step", "evicting"):
        row = roof["interleaved"][name]
        assert row["neighbour_ms"] > 0
        for label in ("default", "all_streaming"):
            assert row[label]["sequence_ms"] > row["neighbour_ms"] and row[label]["caar_ms"] > 0
    sr = roof["subrange_streams"]   # four streams on disjoint quarters of the arrays: ONE cache window per device
    assert sr["streams"] == 4 and sum(sr["elements_per_stream"]) == 1500
    for label in ("device_budget", "per_launch_budget_r03", "all_streaming"):
        assert sr[label]["ms_per_round"] > 0 and 0 < sr[label]["frac"] < 1.2, label
    assert sr["per_launch_budget_r03"]["cache_window_bytes"] == 4 * sr["device_budget"]["cache_window_bytes"]
    rs = roof["run_steps"]   # the driver loop as one launch, beside the per-call headline
    assert rs["one_launch_available"] and rs["calls_per_launch"] == 20
    assert rs["ms_per_call_one_launch"] > 0 and rs["ms_per_call_single_launches"] > 0
    # ... with a roofline of its own: an fp64 rate against the vector peak, no "bandwidth" above the HBM peak
    rr = rs["roofline"]
    assert rr["bound"] == "valu_issue" and rr["unit"] == "TFLOP/s" and 0 < rr["achieved"] < rr["peak"] == 78.6
    assert abs(rr["frac"] - rr["achieved"] / rr["peak"]) < 1e-12 and "algorithmic_GBs_one_launch" not in rs
    # per-launch spread of the single call (median and min), back to back and isolated
    assert 0 < roof["kernel_ms_min"] <= roof["kernel_ms_median"] <= roof["kernel_ms_max"]
    assert 0 < roof["kernel_ms_isolated_min"] <= roof["kernel_ms_isolated_median"]
    import shutil
    if shutil.which("rocprofv3"):   # HBM-side traffic from the counters, measured by the run itself
        assert roof["traffic_source"].startswith("measured in this run"), roof["traffic_source"]
        assert 0.995 <= roof["traffic_over_algorithmic"] <= 1.02, roof["traffic_over_algorithmic"]
        assert abs(roof["traffic"] - roof["traffic_detail"]["read"] - roof["traffic_detail"]["write"]) < 1.0
        # the step-loop kernel's own HBM bytes per call, from the same counter passes: below a single call's, below the peak
        assert 0 < rr["hbm_bytes_per_call"] < roof["traffic"] and 0 < rr["hbm_frac_of_peak"] < 1.0
        assert "steps_kernel" in rr["hbm_kernel"]
    # no figure of the line exceeds a physical peak: every bandwidth-like field is <= the 8 TB/s HBM peak, every
    # fraction-like field <= 1.2 (the Infinity-Cache-assisted algorithmic rates may pass 1.0 of a measured copy, never the peak)
    def walk(o, path=""):
        if isinstance(o, dict):
            for k, v in o.items():
                walk(v, path + "/" + k)
        elif isinstance(o, list):
            for i, v in enumerate(o):
                walk(v, path + "[%d]" % i)
        elif isinstance(o, (int, float)) and not isinstance(o, bool):
            key = path.rsplit("/", 1)[-1].split("[")[0]
            if key.endswith("GBs") or key in ("achieved_all_streaming", "achieved_interleaved", "achieved_interleaved_evicting"):
                assert o <= 8000.0, (path, o)
            if key.startswith("frac") and "copy" not in key and "skeleton" not in key:
                assert o <= 1.0, (path, o)
    walk(j)
    assert roof["dram_estimate"]["frac"] < roof["frac"] <= 1.0
    # the kernel against its own traffic with the arithmetic taken out (same bytes, same cache policy, same arrays)
    assert roof["traffic_skeleton_own_policy_GBs"] > 0 and 0.5 < roof["frac_of_own_traffic_skeleton"] < 1.5
    cb = j["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0
    # SURVEY 8d / BASELINE.md section 4: CPU model, nproc, flags, and a one-core leg on the whole data set (>= 3 timed calls)
    assert cb["cpu_model"] and cb["nproc"] >= cb["cores"] and "-O3" in cb["flags"] or cb["kind"] == "port"
    one = cb["single_core"]
    assert one["elements"] == 1500 and one["calls"] >= 3 and len(one["seconds_per_call"]) == one["calls"] and one["value"] > 0


@pytest.mark.parametrize("launcher", ["plain", "torch.distributed.run"])
def test_two_rank_rehearsal_reports_backend_world_size_and_per_gpu_rates(launcher):
    """`python bench.py --gpus 2` started PLAINLY (bench.py launches its own ranks, self_launch) and under
    torch.distributed.run (the driver's documented command) give the same line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = _clean_env()
    env["CAAR_BENCH_BACKEND"] = "gloo"
    pre = [sys.executable] if launcher == "plain" else [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
        "--master-port", str(port)]
    r = subprocess.run(pre + [BENCH, "--gpus", "2", "--steps", "4", "--warmup", "2", "--elems-per-gpu", "1500",
                              "--no-spinup"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = _line(r.stdout)
    assert j["config"]["launcher"] == ("bench.py" if launcher == "plain" else "torch.distributed.run")
    assert j["config"]["n1_reference"]["workload"] == "NP=4 NLEV=72 num_elems=1500"
    assert j["n_gpus"] == 2 and j["backend"] == "gloo" and j["dist_world_size"] == 2 and j["scaling"] == "weak"
    assert j["warmup_effective"] == 3
    assert "1500 per GPU (3000 total)" in j["config"]["workload"]
    assert abs(j["value"] - 3000 / (j["ms_per_step"] * 1e-3)) <= 1e-6 * j["value"]
    assert j["ms_per_step"] <= j["ms_per_step_incl_closing_barrier"]
    per = j["roofline"]["per_gpu"]
    assert [p["rank"] for p in per] == [0, 1] and all(p["elements"] == 1500 and p["kernel_ms"] > 0 for p in per)
    assert j["roofline"]["kernel_ms"] == max(p["kernel_ms"] for p in per)
    assert "cpu_baseline" not in j and "interleaved" not in j["roofline"]   # N=1 only


def test_six_rank_rehearsal_at_the_configs2_slab_size():
    """The driver's N>1 command line at BASELINE configs[2]'s per-GPU size — `python bench.py --gpus N --steps 5 --warmup 2`,
    12 500 elements per rank, spin-up included — with as many ranks as this pool lets share one GPU.  VERDICT r04 #5 asked for
    N = 8; the pool's process guard admits at most 6 processes on a box's GPU at once (a run with more is killed), so the
    rehearsal is N = 6 (75 000 elements, 14 GB) and the 8-rank tiling / collectives are covered on the CPU
    (tests/test_sharding_gloo.py::test_eight_rank_sharding_gloo, tests/test_bench_launch.py).  Rank coupling is gloo: the
    ranks share the one GPU, so the rates are not scaling figures — the line's shape and the plumbing are what is checked."""
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "6", "--steps", "5", "--warmup", "2"], capture_output=True, text=True,
                       timeout=600, env=dict(_clean_env(), CAAR_BENCH_BACKEND="gloo"))
    wall = time.time() - t0
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 6 and j["backend"] == "gloo" and j["dist_world_size"] == 6 and j["scaling"] == "weak"
    assert j["steps"] == 5 and j["warmup"] == 2 and j["config"]["launcher"] == "bench.py"
    per = j["roofline"]["per_gpu"]
    assert [p["rank"] for p in per] == list(range(6))
    assert [p["elements"] for p in per] == [12500] * 6 and sum(p["elements"] for p in per) == 75000
    assert "12500 per GPU (75000 total)" in j["config"]["workload"]
    assert j["config"]["n1_reference"]["workload"] == "NP=4 NLEV=72 num_elems=12500"
    assert abs(j["value"] - 75000 / (j["ms_per_step"] * 1e-3)) <= 1e-6 * j["value"]
    assert j["roofline"]["kernel_ms"] == max(p["kernel_ms"] for p in per) > 0
    assert wall < 120, wall
    print("6-rank gloo rehearsal: %.1f s wall, %.3g element-updates/s on one shared GPU" % (wall, j["value"]))


def test_a_failing_rank_takes_the_plain_launch_down_with_its_exit_code():
    """Ranks that reject their configuration (NP=5 has no kernel): the plain launch returns non-zero and prints no JSON line."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--np", "5"], capture_output=True, text=True,
                       timeout=300, env=dict(_clean_env(), CAAR_BENCH_BACKEND="gloo"))
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_one_rank_under_rccl_started_plainly():
    """CAAR_BENCH_FORCE_DIST=1 python bench.py --gpus 1: no launcher, no MASTER_PORT in the environment — the one-rank RCCL
    group rendezvouses on 127.0.0.1 and a port bench.py picks."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "4", "--warmup", "2", "--elems-per-gpu", "1500",
                        "--no-spinup", "--no-other-configs", "--no-cpu-baseline", "--no-interleaved", "--no-live-traffic",
                        "--no-steps-leg"], capture_output=True, text=True, timeout=900,
                       env=dict(_clean_env(), CAAR_BENCH_FORCE_DIST="1"))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 1 and j["backend"] == "nccl" and j["dist_world_size"] == 1 and j["config"]["launcher"] == "none"


def test_one_rank_under_rccl_takes_the_collective_path():
    """The N>1 line's plumbing under the backend the driver uses (nccl = RCCL), on the one GPU a test box has: a process group of
    one rank (CAAR_BENCH_FORCE_DIST=1) initialised with device_id, the barriers around the timed region, the MAX all-reduce and
    the per-rank gather on device tensors all run through RCCL."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = _clean_env()
    env["CAAR_BENCH_FORCE_DIST"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "1", "--steps", "4",
                        "--warmup", "2", "--elems-per-gpu", "1500", "--no-spinup", "--no-other-configs", "--no-cpu-baseline",
                        "--no-interleaved", "--no-live-traffic", "--no-steps-leg"], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 1 and j["backend"] == "nccl" and j["dist_world_size"] == 1
    assert abs(j["value"] - 1500 / (j["ms_per_step"] * 1e-3)) <= 1e-6 * j["value"]
    assert j["ms_per_step"] <= j["ms_per_step_incl_closing_barrier"]
    per = j["roofline"]["per_gpu"]
    assert len(per) == 1 and per[0]["rank"] == 0 and per[0]["elements"] == 1500
    assert j["roofline"]["kernel_ms"] == per[0]["kernel_ms"] > 0


def test_performance_lower_bounds():
    """A guard against codegen regressions: the kernels' speed is pinned to one hipcc through a
    17-parameter template, and nothing else in the suite would notice a silent 10 % loss.  Every configuration in a FRESH
    process (tools/perf_guard.py: placement of the arrays moves the rate by 3-5 %, and arrays allocated late in a long-running
    process land badly — measured in this very test: NLEV=128 0.81 in a fresh process, 0.74 after another set had been
    allocated and freed), spun up until stable, best of three blocks of 30 launches, HIP events on the launch stream, the
    adaptive window off (window policy forced).
    Boxes of the pool differ by up to 10 % in what their memory system delivers (round 5: 0.885 on one box, 0.817 on the next
    for the same binary), so each bandwidth-bound figure has two floors and passes if it clears EITHER: the fraction of the
    8 TB/s peak (~4 % under the fast boxes of rounds 4-5, as VERDICT r04 #4 asked) or the rate relative to the same process's
    best tuned device copy (~5 % under what rounds 4-5 measured: 1.08-1.09 / 0.955-0.965 / 1.01 / 0.955), which holds across
    boxes.  The step loop is bound by instruction issue, not by the memory system: one absolute ceiling.
      NP=4 NLEV=72, 10 000 elements   window policy >= 0.84 or 1.03 x copy, all-streaming twin >= 0.74 or 0.91 x copy
      NP=4 NLEV=128, 12 500 elements  >= 0.78 or 0.96 x copy          NP=8 NLEV=72, 20 000 elements  >= 0.74 or 0.90 x copy
      NLEV=72 step loop, 10 000 elements, 20 calls per launch  <= 0.115 ms per call."""
    got = {}
    env = _clean_env()
    env.pop("CAAR_PLACEMENT_POOL_GIB", None)   # (tests/conftest.py shrinks the placement pool for speed; here the library's default counts)
    for np_, nlev, elems, extra in ((4, 72, 10000, ["--twin", "--steps"]), (4, 128, 12500, []), (8, 72, 20000, [])):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "perf_guard.py"), "--np", str(np_), "--nlev", str(nlev),
                            "--elems", str(elems)] + extra, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
        got["np%d_nlev%d" % (np_, nlev)] = _line(r.stdout)
    print("performance guard:", json.dumps(got))
    a, b, c = got["np4_nlev72"], got["np4_nlev128"], got["np8_nlev72"]
    assert a["frac"] >= 0.84 or a["over_copy"] >= 1.03, a
    assert a["all_streaming_frac"] >= 0.74 or a["all_streaming_over_copy"] >= 0.91, a
    assert a["step_loop_ms_per_call"] <= 0.115, a
    assert b["frac"] >= 0.78 or b["over_copy"] >= 0.96, b
    assert c["frac"] >= 0.74 or c["over_copy"] >= 0.90, c
    # ... and floors no box has come near, whatever its copy rate
    assert a["frac"] >= 0.72 and b["frac"] >= 0.68 and c["frac"] >= 0.64, got
