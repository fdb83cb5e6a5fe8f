"""How a host uses the path beyond one call: time stepping with rotating levels, an empty
element range, extra tracer slots, side streams and hipGraph capture."""
import os
import sys

import numpy as np
import pytest
import torch

import cases
from oracle import pyoracle as po

import tinman_sandbox_amd as tsa

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_time_stepping_trajectory_matches_oracle(oracle):
    """Five calls with update_time_levels between them (TestData::update_time_levels,
    data_structures.cpp:174-180; the reference driver has the call commented out,
    main.cpp:118, so this is the real leap-frog use).  Rounding differences feed back
    through the state, so the bound is looser than for one call."""
    arrs, Dvv, sc = cases.make_case("np4_nlev72_hashed")
    sc["dt2"] = 0.5
    ref = cases.copy_arrays(arrs)
    data = tsa.TestData.from_numpy(arrs, Dvv, sc, device="cuda")
    s = dict(sc)
    for _ in range(5):
        oracle.compute_and_apply_rhs(ref, Dvv, s)
        s["np1"], s["nm1"], s["n0"] = s["nm1"], s["n0"], s["np1"]
        tsa.compute_and_apply_rhs(data)
        data.update_time_levels()
    torch.cuda.synchronize()
    assert (data.control.n0, data.control.np1, data.control.nm1) == (s["n0"], s["np1"], s["nm1"])
    got = data.arrays.to_numpy()
    for n in tsa.caar.MUTATED:
        assert cases.scaled_err(got[n], ref[n]) <= 1e-11, n


def test_empty_element_range_is_a_noop():
    arrs, Dvv, sc = cases.make_case("np4_nlev72_closed_dry")
    sc["nets"] = sc["nete"] = 1
    data = tsa.TestData.from_numpy(arrs, Dvv, sc, device="cuda")
    tsa.compute_and_apply_rhs(data)
    torch.cuda.synchronize()
    got = data.arrays.to_numpy()
    for n in tsa.ARRAY_NAMES:
        assert np.array_equal(got[n], arrs[n]), n


def test_extra_tracer_slots_are_ignored(oracle):
    """qsize_d = 3: the path reads tracer 0 only (P:143 SLICE_6D_IJK(..., ie, 0, qn0, ...))."""
    arrs = cases.hashed_arrays(4, 72, 2, seed=31, qsize_d=3)
    sc = po.default_scalars(72)
    sc.update(qn0=1, dt2=3.0)
    Dvv = cases.dvv_for(4)
    want = cases.copy_arrays(arrs)
    oracle.compute_and_apply_rhs(want, Dvv, sc)
    data = tsa.TestData.from_numpy(arrs, Dvv, sc, device="cuda")
    tsa.compute_and_apply_rhs(data)
    torch.cuda.synchronize()
    got = data.arrays.to_numpy()
    for n in tsa.caar.MUTATED:
        assert cases.scaled_err(got[n], want[n]) <= 1e-12, n
    assert np.array_equal(got["elem_state_Qdp"], arrs["elem_state_Qdp"])


def test_side_stream_and_graph_capture(oracle):
    """caar_launch allocates nothing and never synchronises: it runs on any stream and can
    be captured into a hipGraph and replayed."""
    arrs, Dvv, sc = cases.make_case("np4_nlev72_closed")
    want = cases.copy_arrays(arrs)
    oracle.compute_and_apply_rhs(want, Dvv, sc)
    data = tsa.TestData.from_numpy(arrs, Dvv, sc, device="cuda")
    data.dvv_device()  # upload Dvv outside the capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        tsa.compute_and_apply_rhs(data, side)
    side.synchronize()
    first = data.arrays.to_numpy()
    for n in ("elem_state_v", "elem_state_T", "elem_state_dp3d", "elem_derived_phi"):
        assert cases.scaled_err(first[n], want[n]) <= 1e-12, n

    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        tsa.compute_and_apply_rhs(data)  # captured on the capture stream
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    again = data.arrays.to_numpy()
    # the np1 state is idempotent; the accumulators have now received 1 + capture(0) + 2 updates
    for n in ("elem_state_v", "elem_state_T", "elem_state_dp3d", "elem_derived_phi"):
        assert np.array_equal(again[n], first[n]), n
    inc = first["elem_derived_vn0"] - arrs["elem_derived_vn0"]
    total = again["elem_derived_vn0"] - arrs["elem_derived_vn0"]
    assert cases.scaled_err(total, 3 * inc) <= 1e-12


def test_single_element_and_large_offsets(oracle):
    """One element at the far end of a long array: 64-bit offsets, last workgroup."""
    E = 4001
    data = tsa.TestData().init_data(E, 4, 72, device="cuda")
    data.control.nets, data.control.nete = E - 1, E
    before = {n: data.arrays[n].clone() for n in tsa.caar.MUTATED}
    tsa.compute_and_apply_rhs(data)
    torch.cuda.synchronize()
    O = oracle
    ref = O.init_arrays(4, 72, 1, 3, E)
    sc = po.default_scalars(72)
    sc["nets"], sc["nete"] = E - 1, E
    O.compute_and_apply_rhs(ref, O.dvv_np4(False), sc)
    for n in tsa.caar.MUTATED:
        got = data.arrays[n].cpu().numpy()
        assert cases.scaled_err(got[E - 1], ref[n][E - 1]) <= 1e-12, n
        assert torch.equal(data.arrays[n][: E - 1], before[n][: E - 1]), n


def test_eta_noop_update_is_bit_faithful(oracle):
    """P:172,181: eta_dot_dpdn += eta_ave_w * 0.  Arithmetically a no-op, but -0.0 becomes +0.0
    and the reference does perform the read-modify-write; the default kernel does too, and
    the variant that stores only changed values must end with the same bits."""
    lib = tsa.library().lib
    arrs, Dvv, sc = cases.make_case("np4_nlev72_hashed")
    eta = arrs["elem_derived_eta_dot_dpdn"]
    eta[1, ::7] = -0.0
    eta[2, 3] = 0.0
    want = cases.copy_arrays(arrs)
    oracle.compute_and_apply_rhs(want, Dvv, sc)
    w = want["elem_derived_eta_dot_dpdn"]
    assert not np.signbit(w[1, ::7]).any() and np.signbit(arrs["elem_derived_eta_dot_dpdn"][1, ::7]).all()
    names = [lib.caar_variant_info(4, 72, v).decode() for v in range(lib.caar_num_variants(4, 72))]
    cond = [i for i, n in enumerate(names) if "bits change" in n]
    assert cond, names
    try:
        for v in [0] + cond:
            lib.caar_select_variant(4, 72, v)
            data = tsa.TestData.from_numpy(arrs, Dvv, sc, device="cuda")
            tsa.compute_and_apply_rhs(data)
            torch.cuda.synchronize()
            got = data.arrays["elem_derived_eta_dot_dpdn"].cpu().numpy()
            assert np.array_equal(got.view(np.int64), w.view(np.int64)), v  # bit for bit, incl. the sign of zero
    finally:
        lib.caar_select_variant(4, 72, 0)


def test_device_reciprocal_is_within_one_ulp():
    """The kernels replace x/p by x*recip(p) (DESIGN.md "Numerics"): recip must be <= 1 ulp
    from the correctly rounded 1/p over the whole range pressures can take."""
    import ctypes as C
    L = tsa.library()
    n = 1 << 20
    x = np.exp(cases.uniform((n,), 71, np.log(1e-6), np.log(1e9)))
    x[:4] = [1.0, 2.0, 3.0, 1e5]
    xd = torch.from_numpy(x).cuda()
    out = torch.empty_like(xd)
    st = torch.cuda.current_stream()
    L.check(L.lib.caar_reciprocal(C.c_void_p(xd.data_ptr()), C.c_void_p(out.data_ptr()), n,
                                  C.c_void_p(st.cuda_stream)), "caar_reciprocal")
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    want = 1.0 / x
    ulp = np.spacing(want)
    assert np.max(np.abs(got - want) / ulp) <= 1.0
    assert got[0] == 1.0 and got[1] == 0.5


def test_four_time_levels(oracle):
    """The layout's time-level count is a run-time dimension (the reference fixes 3,
    config.h.in:7): four levels, path on levels (3, 0, 2)."""
    arrs = cases.hashed_arrays(4, 72, 2, seed=81, timelevels=4)
    sc = po.default_scalars(72)
    sc.update(n0=3, np1=0, nm1=2, dt2=2.0)
    Dvv = cases.dvv_for(4)
    want = cases.copy_arrays(arrs)
    oracle.compute_and_apply_rhs(want, Dvv, sc)
    data = tsa.TestData.from_numpy(arrs, Dvv, sc, device="cuda")
    assert data.arrays.timelevels == 4
    tsa.compute_and_apply_rhs(data)
    torch.cuda.synchronize()
    got = data.arrays.to_numpy()
    for n in tsa.caar.MUTATED:
        assert cases.scaled_err(got[n], want[n]) <= 1e-12, n
    for t in (1, 2, 3):
        assert np.array_equal(got["elem_state_T"][:, t], arrs["elem_state_T"][:, t])


@pytest.mark.parametrize("np_,nlev,E", [(4, 72, 100000), (4, 128, 100000), (8, 72, 40000)])
def test_whole_job_on_one_gpu_beyond_4gib_offsets(oracle, np_, nlev, E):
    """BASELINE configs[2]/[3]'s global element count on ONE GPU (18.6 / 33 GB resident, NP=8:
    30 GB): single arrays exceed 4 GiB, so element offsets must be 64-bit everywhere.  Elements
    on both sides of the 4 GiB mark of state_v and the last one are checked against the oracle
    run on copies of exactly those elements' inputs."""
    data = tsa.TestData().init_data(E, np_, nlev, device="cuda")
    data.control.qn0, data.control.dt2 = 0, 0.5
    per_elem_v = data.arrays["elem_state_v"][0].numel() * 8
    mark = (1 << 32) // per_elem_v
    assert mark + 1 < E
    picks = [0, mark - 1, mark, mark + 1, E // 2, E - 1]
    sub = {n: data.arrays[n][picks].cpu().numpy().copy() for n in tsa.ARRAY_NAMES}
    tsa.compute_and_apply_rhs(data)
    torch.cuda.synchronize()
    sc = po.default_scalars(nlev)
    sc.update(qn0=0, dt2=0.5)
    oracle.compute_and_apply_rhs(sub, data.deriv.Dvv, sc)
    for n in tsa.caar.MUTATED:
        got = data.arrays[n][picks].cpu().numpy()
        assert cases.scaled_err(got, sub[n]) <= 1e-12, n
    # every element was updated: no np1 value is left at its initial closed form
    T1 = data.arrays["elem_state_T"][:, 1]
    assert bool(torch.isfinite(T1).all())
    # the on-device print_results_2norm over the whole range, against plain sums of squares
    nv, nT, ndp = tsa.state_norms(data)
    for got_norm, name in ((nv, "elem_state_v"), (nT, "elem_state_T"), (ndp, "elem_state_dp3d")):
        x = data.arrays[name][:, 1]
        assert abs(got_norm - float(torch.sqrt((x * x).sum()))) <= 1e-12 * got_norm, name
    del data
    torch.cuda.empty_cache()


def test_disjoint_element_ranges_on_concurrent_streams(oracle):
    """HOMME's horizontal threading: several callers, each with its own Control{nets, nete}
    (data_structures.hpp:58-69), on disjoint element ranges of the same arrays.  Three
    streams at once must give bit for bit what one launch over the whole range gives."""
    arrs = cases.hashed_arrays(4, 72, 300, seed=211)
    Dvv = cases.dvv_for(4)
    sc = po.default_scalars(72)
    sc.update(qn0=1, dt2=0.5)
    whole = tsa.TestData.from_numpy(arrs, Dvv, sc, device="cuda")
    tsa.compute_and_apply_rhs(whole)
    torch.cuda.synchronize()
    want = whole.arrays.to_numpy()

    data = tsa.TestData.from_numpy(arrs, Dvv, sc, device="cuda")
    data.dvv_device()
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    cuts = [0, 77, 201, 300]
    for rep in range(1):
        for s, (a, b) in zip(streams, zip(cuts[:-1], cuts[1:])):
            data.control.nets, data.control.nete = a, b
            tsa.compute_and_apply_rhs(data, s)
    torch.cuda.synchronize()
    got = data.arrays.to_numpy()
    for n in tsa.ARRAY_NAMES:
        assert np.array_equal(got[n], want[n]), n


def test_misaligned_arrays_are_refused():
    """v and vn0 move as 16-byte (u, v) pairs: a base that is only 8-byte aligned is an error,
    not a fault."""
    import ctypes as C
    from tinman_sandbox_amd import caar as m
    L = tsa.library()
    data = tsa.TestData().init_data(4, 4, 72, device="cuda")
    dims, ptrs, prm = data.arrays.dims(), data.arrays.pointers(), data.params()
    shifted = C.cast(C.c_void_p(data.arrays["elem_state_v"].data_ptr() + 8), m._dp)
    ptrs.elem_state_v = shifted
    rc = L.lib.caar_launch(C.byref(dims), C.byref(ptrs), C.c_void_p(data.dvv_device().data_ptr()), C.byref(prm), None)
    assert rc == -1 and b"invalid" in L.lib.caar_strerror(rc)


def test_half_the_hbm_in_one_launch(oracle):
    """700 000 elements (130 GB resident, NP=4 NLEV=72) in one launch: state_v alone holds
    4.8e9 doubles, so element offsets exceed 2^32 DOUBLES, not just 2^32 bytes.  Elements on
    both sides of that mark and the last one are checked against the oracle."""
    E = 700000
    free, _ = torch.cuda.mem_get_info()
    if free < 200 * 2 ** 30:
        pytest.skip("needs 200 GiB of free HBM (array initialisation temporaries included)")
    data = tsa.TestData().init_data(E, 4, 72, device="cuda")
    data.control.qn0, data.control.dt2 = 0, 0.5
    per_elem_v = data.arrays["elem_state_v"][0].numel()
    mark = (1 << 32) // per_elem_v
    assert mark + 1 < E
    picks = [0, mark - 1, mark, mark + 1, E - 1]
    sub = {n: data.arrays[n][picks].cpu().numpy().copy() for n in tsa.ARRAY_NAMES}
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    tsa.compute_and_apply_rhs(data)
    b.record()
    torch.cuda.synchronize()
    sc = po.default_scalars(72)
    sc.update(qn0=0, dt2=0.5)
    oracle.compute_and_apply_rhs(sub, data.deriv.Dvv, sc)
    for n in tsa.caar.MUTATED:
        got = data.arrays[n][picks].cpu().numpy()
        assert cases.scaled_err(got, sub[n]) <= 1e-12, n
    print("700000 elements: %.2f ms, %.0f GB/s algorithmic" % (
        a.elapsed_time(b), tsa.algorithmic_bytes(4, 72) * E / a.elapsed_time(b) / 1e6))
    del data
    torch.cuda.empty_cache()


@pytest.mark.parametrize("nlev,rsplit", [(72, 1), (72, 0), (128, 1), (128, 0)])
def test_cache_window_changes_nothing_but_speed(nlev, rsplit):
    """caar_set_cache_window: which elements keep their accumulators in the Infinity Cache is a
    cache-policy choice; every setting must give bit-identical arrays — in the Lagrangian and the Eulerian form
    (both run the hybrid policy by default), windows that keep none, some and all of the elements."""
    lib = tsa.library().lib
    arrs = cases.hashed_arrays(4, nlev, 37, seed=251 + nlev)
    Dvv = cases.dvv_for(4)
    sc = po.default_scalars(nlev)
    sc.update(qn0=1, dt2=0.5, nets=2, nete=35)
    if rsplit == 0:
        sc.update(rsplit=0, hybi=(np.arange(nlev + 1) / nlev) ** 2)
    ref = None
    try:
        for window in (192 << 20, 0, 1 << 16, 300 * 1024, 700 * 1024, 1 << 40):
            assert lib.caar_set_cache_window(window) == 0
            data = tsa.TestData.from_numpy(arrs, Dvv, sc, device="cuda")
            tsa.compute_and_apply_rhs(data)
            tsa.compute_and_apply_rhs(data)
            torch.cuda.synchronize()
            got = data.arrays.to_numpy()
            if ref is None:
                ref = got
            for n in tsa.ARRAY_NAMES:
                assert np.array_equal(got[n], ref[n]), (window, n)
        assert lib.caar_set_cache_window(-1) != 0
    finally:
        lib.caar_set_cache_window(224 << 20)


def test_arrays_placed_for_bandwidth_compute_the_same(oracle, monkeypatch):
    """caar_arrays_alloc[_ex] (include/caar.h; DESIGN.md section 5 "Placement"): arrays backed by physical chunks spread over
    the device's address classes through HIP virtual memory management.  Same results bit for bit as torch-allocated
    arrays, chunk-backed from 256 MiB on, plain hipMalloc below that or on request (per call: CaarPlacement; or for a
    whole process: CAAR_PLACEMENT), memory returned when released."""
    import gc
    E = 2000  # 372 MB of arrays
    spread = tsa.TestData().init_data(E, 4, 72, device="cuda")
    assert spread.arrays.arena is not None and spread.arrays.arena.spread()
    for n in tsa.ARRAY_NAMES:
        assert spread.arrays[n].data_ptr() % (2 << 20) == 0, n      # every array starts on a chunk boundary of the range
    plain = tsa.TestData().init_data(E, 4, 72, device="cuda", place="torch")
    assert plain.arrays.arena is None
    for n in tsa.ARRAY_NAMES:
        assert torch.equal(spread.arrays[n], plain.arrays[n]), n     # same initial data
    for _ in range(2):
        tsa.compute_and_apply_rhs(spread)
        tsa.compute_and_apply_rhs(plain)
    torch.cuda.synchronize()
    for n in tsa.ARRAY_NAMES:
        assert torch.equal(spread.arrays[n], plain.arrays[n]), n
    # against the oracle on a few elements
    arrs = {n: spread.arrays[n][:3].cpu().numpy().copy() for n in tsa.ARRAY_NAMES}
    want = oracle.init_arrays(4, 72, 1, 3, 3)
    sc = po.default_scalars(72)
    for _ in range(2):
        oracle.compute_and_apply_rhs(want, oracle.dvv_np4(False), sc)
    for n in cases.OUTPUT_NAMES:
        assert cases.scaled_err(arrs[n], want[n]) <= 1e-12, n
    # small data sets and CAAR_PLACEMENT=malloc: plain allocations through the same entry point
    small = tsa.TestData().init_data(8, 4, 72, device="cuda")
    assert small.arrays.arena is not None and not small.arrays.arena.spread()
    monkeypatch.setenv("CAAR_PLACEMENT", "malloc")
    forced = tsa.TestData().init_data(E, 4, 72, device="cuda")
    assert forced.arrays.arena is not None and not forced.arrays.arena.spread()
    # the per-call choice wins over the environment, both ways
    chosen = tsa.TestData().init_data(E, 4, 72, device="cuda", place=tsa.placement("spread", pool_gib=4))
    assert chosen.arrays.arena.spread() and chosen.arrays.arena.pool_bytes() <= 4 << 30
    monkeypatch.delenv("CAAR_PLACEMENT")
    forced2 = tsa.TestData().init_data(E, 4, 72, device="cuda", place=tsa.placement("malloc"))
    assert forced2.arrays.arena is not None and not forced2.arrays.arena.spread()
    # the pool is bounded by the share of the FREE memory the caller allows
    free_now = torch.cuda.mem_get_info()[0]
    tight = tsa.TestData().init_data(E, 4, 72, device="cuda", place=tsa.placement("spread", pool_gib=1024, max_free_fraction=0.05))
    assert tight.arrays.arena.spread() and tight.arrays.arena.pool_bytes() <= 0.05 * free_now + (64 << 20)
    del chosen, forced2, tight
    # released memory comes back (tensors keep the arena alive until the last one is gone)
    del spread, plain, small, forced, arrs
    gc.collect()
    torch.cuda.empty_cache()
    def cycle():
        d = tsa.TestData().init_data(E, 4, 72, device="cuda")
        keep = d.arrays["elem_state_T"]
        del d
        gc.collect()
        assert keep.sum().item() != 0.0   # still valid: the view holds the arena
        del keep
        gc.collect()
        torch.cuda.empty_cache()
        return torch.cuda.mem_get_info()[0]
    cycle()
    free1 = cycle()
    for _ in range(4):
        free2 = cycle()
    assert free1 - free2 < (64 << 20), (free1, free2)   # no physical memory lost per allocate / release cycle


def test_arenas_allocated_and_released_in_turn_stay_correct(oracle):
    """The sequence that gave wrong results in round 2 — arenas of different sizes allocated, run, checked and released in
    turn.  On ROCm 7.2 / gfx950 a virtual address that is mapped a second time keeps translating to its first physical
    memory (bare-HIP reproducer tools/probes/vmm_va_reuse_probe.hip, profiles/r03/vmm_va_reuse_probe.log), so the
    allocator unmaps chunk by chunk, returns the physical memory and keeps a released arena's range reserved: no address is
    ever mapped twice, and every new arena must see ITS memory."""
    import gc
    for rep in range(2):
        for np_, nlev, E, gold_name in ((4, 72, 6000, "np4_nlev72_closed"), (4, 128, 3000, "np4_nlev128_closed"),
                                        (8, 72, 1500, "np8_nlev72_closed"), (4, 72, 2500, "np4_nlev72_closed")):
            data = tsa.TestData().init_data(E, np_, nlev, device="cuda")
            assert data.arrays.arena.spread()
            tsa.compute_and_apply_rhs(data)
            torch.cuda.synchronize()
            gold = cases.load_golden(gold_name)
            ng = gold["elem_derived_phi"].shape[0]
            sc = po.default_scalars(nlev)
            for n in cases.OUTPUT_NAMES:
                g = data.arrays[n][:ng].cpu().numpy()
                g = g[:, sc["np1"]] if n.startswith("elem_state_") else g
                assert cases.scaled_err(g, gold[n]) <= 1e-12, (rep, np_, nlev, E, n)
            # the tail of the arrays as well (the last chunks of the range): against the same launch on torch's allocations
            other = tsa.TestData().init_data(E, np_, nlev, device="cuda", place="torch")
            tsa.compute_and_apply_rhs(other)
            torch.cuda.synchronize()
            for n in tsa.ARRAY_NAMES:
                assert torch.equal(data.arrays[n], other.arrays[n]), (rep, np_, nlev, E, n)
            del data, other
            gc.collect()
            torch.cuda.empty_cache()


_TWO_PROC_SCRIPT = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch
import tinman_sandbox_amd as tsa
import cases
from oracle import pyoracle as po   # checker only
me, flag_dir = sys.argv[2], sys.argv[3]
torch.cuda.init()
open(os.path.join(flag_dir, "ready_" + me), "w").close()
t0 = time.time()
while not (os.path.exists(os.path.join(flag_dir, "ready_a")) and os.path.exists(os.path.join(flag_dir, "ready_b"))):
    assert time.time() - t0 < 300, "the other process never came up"
    time.sleep(0.005)
t1 = time.time()
data = tsa.TestData().init_data(10000, 4, 72, device="cuda")   # library default placement, default pool
t2 = time.time()
for _ in range(2):
    tsa.compute_and_apply_rhs(data)
torch.cuda.synchronize()
O = po.Oracle()
want = O.init_arrays(4, 72, 1, 3, 3)
sc = po.default_scalars(72)
for _ in range(2):
    O.compute_and_apply_rhs(want, O.dvv_np4(False), sc)
worst = 0.0
for n in cases.OUTPUT_NAMES:
    got = data.arrays[n][:3].cpu().numpy()
    worst = max(worst, cases.scaled_err(got, want[n]))
print("RESULT", me, "spread", int(data.arrays.arena.spread()), "pool_gib", data.arrays.arena.pool_bytes() / 2**30,
      "alloc_s", round(t2 - t1, 2), "worst", worst, flush=True)
assert worst <= 1e-12
"""


def test_two_processes_allocate_on_one_gpu_at_the_same_time(tmp_path):
    """Two ranks sharing a GPU (normal for E3SM hosts) create 10 000-element data sets at the same moment with the
    library's default placement: the temporary pools are bounded by half of what is free when each call starts, and a
    pool cut short by the neighbour is not an error — both succeed and match the oracle."""
    import subprocess
    script = tmp_path / "two_proc.py"
    script.write_text(_TWO_PROC_SCRIPT)
    env = dict(os.environ)
    env.pop("CAAR_PLACEMENT_POOL_GIB", None)   # the library's own default pool, not the suite's small one
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, me, str(tmp_path)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True, env=env) for me in ("a", "b")]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-2000:]
        assert "RESULT" in o, o[-2000:]
    print("\n".join(l for o in outs for l in o.splitlines() if l.startswith("RESULT")))


def test_context_api_allocates_through_the_placed_allocator(oracle):
    """caar_create backs its arrays the same way; results as before."""
    import ctypes as C
    from tinman_sandbox_amd import caar as m
    L = tsa.library()
    E = 1500
    arrs = oracle.init_arrays(4, 72, 1, 3, E)
    want = cases.copy_arrays(arrs)
    sc = po.default_scalars(72)
    Dvv = oracle.dvv_np4(False)
    sc2 = dict(sc, nets=0, nete=4)
    oracle.compute_and_apply_rhs(want, Dvv, sc2)
    dims = m._CaarDims(4, 72, 1, 3, E)
    ctx = C.c_void_p()
    L.check(L.lib.caar_create(C.byref(ctx), C.byref(dims), 0), "create")
    host = m._CaarArrays(*[arrs[n].ctypes.data_as(m._dp) for n in tsa.ARRAY_NAMES])
    L.check(L.lib.caar_upload(ctx, C.byref(host), 0, E), "upload")
    dvv = np.ascontiguousarray(Dvv)
    prm = m._CaarParams(0, 4, sc["n0"], sc["np1"], sc["nm1"], sc["qn0"], sc["dt2"], sc["rrearth"], sc["eta_ave_w"],
                        sc["Rwater_vapor"], sc["Rgas"], sc["kappa"], sc["ps0"], float(sc["hyai"][0]),
                        dvv.ctypes.data_as(m._dp), 1, None, None)
    L.check(L.lib.caar_run(ctx, C.byref(prm)), "run")
    L.check(L.lib.caar_download(ctx, C.byref(host), 0, E, 0), "download")
    L.check(L.lib.caar_sync(ctx), "sync")
    L.lib.caar_destroy(ctx)
    for n in cases.OUTPUT_NAMES:
        assert cases.scaled_err(arrs[n][:4], want[n][:4]) <= 1e-12, n
        assert np.array_equal(arrs[n][4:], want[n][4:]), n


@pytest.mark.parametrize("np_,nlev,rsplit,knob", [(8, 72, 1, 0), (4, 72, 1, 0), (4, 30, 1, 1), (4, 72, 0, 1), (4, 100, 1, 1)])
def test_launch_steps_without_a_step_loop_kernel_equals_single_calls(np_, nlev, rsplit, knob):
    """caar_launch_steps on its host-side fallback — nsteps launches of caar_launch with the rotation in between — where no
    fused kernel exists (level counts without one, the Eulerian form) and where one exists but is switched off
    (caar_set_fused_steps(0): NP=8 and NP=4 NLEV=72): bitwise what the host's own loop gives, Control rotated nsteps times."""
    lib = tsa.library().lib
    if knob:   # these configurations must not have a step loop of their own for the case to mean what it says
        assert rsplit == 0 or not any(lib.caar_has_fused_steps(np_, nlev, v) for v in range(lib.caar_num_variants(np_, nlev)))
    lib.caar_set_fused_steps(knob)
    try:
        _steps_fallback_case(np_, nlev, rsplit)
    finally:
        lib.caar_set_fused_steps(1)


def test_a_single_step_takes_the_tuned_single_launch():
    """nsteps == 1 never goes to a step-loop kernel (the lone call is what the hybrid-policy single launch is tuned for):
    same bits either way, and under rocprofv3 the kernel name says which ran; here: the result equals caar_launch's."""
    arrs = cases.hashed_arrays(4, 72, 6, seed=77)
    sc = po.default_scalars(72)
    sc.update(dt2=0.5, qn0=1)
    a = tsa.TestData.from_numpy(arrs, cases.dvv_for(4), sc, device="cuda")
    b = tsa.TestData.from_numpy(arrs, cases.dvv_for(4), sc, device="cuda")
    tsa.compute_and_apply_rhs_steps(a, 1, True)
    tsa.compute_and_apply_rhs(b)
    torch.cuda.synchronize()
    for n in tsa.ARRAY_NAMES:
        assert torch.equal(a.arrays[n], b.arrays[n]), n


def _steps_fallback_case(np_, nlev, rsplit):
    arrs = cases.hashed_arrays(np_, nlev, 5, seed=400 + np_ + nlev)
    Dvv = cases.dvv_for(np_)
    sc = po.default_scalars(nlev)
    sc.update(dt2=0.125, qn0=1, nets=1, nete=4)
    if rsplit == 0:
        sc.update(rsplit=0, hybi=(np.arange(nlev + 1) / nlev) ** 2)
    a = tsa.TestData.from_numpy(arrs, Dvv, sc, device="cuda")
    b = tsa.TestData.from_numpy(arrs, Dvv, sc, device="cuda")
    tsa.compute_and_apply_rhs_steps(a, 4, True)
    for _ in range(4):
        tsa.compute_and_apply_rhs(b)
        b.update_time_levels()
    torch.cuda.synchronize()
    assert (a.control.n0, a.control.np1, a.control.nm1) == (b.control.n0, b.control.np1, b.control.nm1)
    for n in tsa.ARRAY_NAMES:
        assert torch.equal(a.arrays[n], b.arrays[n]), n
    L = tsa.library()
    dims, ptrs, prm = a.arrays.dims(), a.arrays.pointers(), a.params(device_constants=True)
    import ctypes as C
    assert L.lib.caar_launch_steps(C.byref(dims), C.byref(ptrs), C.c_void_p(a.dvv_device().data_ptr()), C.byref(prm), 0, 1,
                                   None) == -1   # nsteps < 1


@pytest.mark.parametrize("np_,nlev,E", [(4, 72, 300), (4, 72, 900), (4, 128, 200), (8, 72, 60)])
def test_long_step_loop_with_extra_tracers_and_time_levels(np_, nlev, E):
    """25 calls in one launch on arrays with 3 tracer slots and 4 time levels (the reference's dimensions are compile-time
    constants, config.h.in: QSIZE_D, NUM_TIME_LEVELS), second tracer slot, a sub-range of the elements — bitwise what 25
    single launches leave (E = 300 / 900: both NLEV=72 step loops, the all-on-chip one and the two-workgroup one)."""
    lib = tsa.library().lib
    arrs = cases.hashed_arrays(np_, nlev, 4, seed=900 + np_ + nlev, qsize_d=3, timelevels=4)
    reps = -(-E // 4)
    arrs = {k: np.concatenate([v] * reps, axis=0)[:E].copy() for k, v in arrs.items()}
    Dvv = cases.dvv_for(np_)
    sc = po.default_scalars(nlev)
    sc.update(dt2=1.0e-3, eta_ave_w=0.25, qn0=1, n0=3, np1=0, nm1=2, nets=5, nete=E - 7)
    a = tsa.TestData.from_numpy(arrs, Dvv, sc, device="cuda")
    b = tsa.TestData.from_numpy(arrs, Dvv, sc, device="cuda")
    try:
        tsa.compute_and_apply_rhs_steps(a, 25, True)
        lib.caar_set_fused_steps(0)
        tsa.compute_and_apply_rhs_steps(b, 25, True)
    finally:
        lib.caar_set_fused_steps(1)
    torch.cuda.synchronize()
    for n in tsa.ARRAY_NAMES:
        assert torch.equal(a.arrays[n], b.arrays[n]), n
        assert torch.isfinite(a.arrays[n]).all(), n
    # nothing outside [nets, nete) and no untouched time level / tracer slot changed
    for n in tsa.caar.MUTATED:
        assert np.array_equal(a.arrays[n][:5].cpu().numpy(), arrs[n][:5]), n
        assert np.array_equal(a.arrays[n][E - 7:].cpu().numpy(), arrs[n][E - 7:]), n
    assert np.array_equal(a.arrays["elem_state_T"][:, 1].cpu().numpy(), arrs["elem_state_T"][:, 1])   # level 1 is never used
    assert np.array_equal(a.arrays["elem_state_Qdp"].cpu().numpy(), arrs["elem_state_Qdp"])


def test_placement_arguments_are_validated():
    import ctypes as C
    from tinman_sandbox_amd import caar as m
    L = tsa.library()
    dims = m._CaarDims(4, 72, 1, 3, 100)
    out = m._CaarArrays()
    for bad in (m._CaarPlacement(7, 0, 0.0), m._CaarPlacement(-1, 0, 0.0), m._CaarPlacement(1, -5, 0.0),
                m._CaarPlacement(1, 0, 1.5), m._CaarPlacement(1, 0, -0.1)):
        h = C.c_void_p()
        assert L.lib.caar_arrays_alloc_ex(C.byref(h), C.byref(dims), 0, C.byref(bad), C.byref(out)) == -1
        ctx = C.c_void_p()
        assert L.lib.caar_create_ex(C.byref(ctx), C.byref(dims), 0, C.byref(bad)) == -1
    h = C.c_void_p()
    assert L.lib.caar_arrays_alloc_ex(C.byref(h), C.byref(dims), 99, None, C.byref(out)) == -1   # no such device
    assert L.lib.caar_arrays_free(None) == -1


_DEBUG_SCRIPT = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch
import tinman_sandbox_amd as tsa
import cases
from oracle import pyoracle as po   # checker only
lib = tsa.library().lib
assert tsa.caar.LIB_PATH.endswith("libcaar_hip_debug.so")
total = 0
for name in ("np4_nlev72_closed", "np4_nlev128_closed", "np8_nlev72_closed"):
    arrs, Dvv, sc = cases.make_case(name)
    # healthy state: nothing to report, same results as ever
    want = cases.copy_arrays(arrs)
    po.Oracle().compute_and_apply_rhs(want, Dvv, sc)
    data = tsa.TestData.from_numpy(arrs, Dvv, sc, device="cuda")
    tsa.compute_and_apply_rhs(data)
    assert lib.caar_debug_dp3d_violations(0) == total, name
    got = data.arrays.to_numpy()
    for n in cases.OUTPUT_NAMES:
        assert cases.scaled_err(got[n], want[n]) <= 1e-12, (name, n)
    # a layer thickness that goes non-positive at np1 (dp3d(nm1) far below zero in a few places)
    bad = cases.copy_arrays(arrs)
    bad["elem_state_dp3d"][0, sc["nm1"], 3:5] = -1.0e6
    want = cases.copy_arrays(bad)
    po.Oracle().compute_and_apply_rhs(want, Dvv, sc)
    expect = int((want["elem_state_dp3d"][sc["nets"]:sc["nete"], sc["np1"]] <= 0).sum())
    assert expect > 0
    data = tsa.TestData.from_numpy(bad, Dvv, sc, device="cuda")
    tsa.compute_and_apply_rhs(data)
    total += expect
    assert lib.caar_debug_dp3d_violations(0) == total, (name, lib.caar_debug_dp3d_violations(0), total)
# the step-loop kernel has its own counter
arrs, Dvv, sc = cases.make_case("np4_nlev72_closed")
arrs["elem_state_dp3d"][1, sc["nm1"], 7] = -1.0e6
data = tsa.TestData.from_numpy(arrs, Dvv, sc, device="cuda")
tsa.compute_and_apply_rhs_steps(data, 1, True)
assert lib.caar_debug_dp3d_violations(1) > total
assert lib.caar_debug_dp3d_violations(0) == 0     # reset
print("DEBUG-BUILD-OK", total)
"""


def test_debug_build_counts_nonpositive_layer_thickness(tmp_path):
    """-DCAAR_DEBUG (libcaar_hip_debug.so): the reference's only hot-path assertion, check_dp3d
    (level_vectorized_ppscan/CaarFunctor.hpp:82-97: dp3d(np1) > 0), as a device-side counter — as many violations as the
    oracle's result has non-positive dp3d(np1) entries, none on healthy data, same results; the release build says -1."""
    import subprocess
    assert tsa.library().lib.caar_debug_dp3d_violations(0) == -1
    script = tmp_path / "debug_build.py"
    script.write_text(_DEBUG_SCRIPT)
    env = dict(os.environ, CAAR_LIBRARY="debug")
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "DEBUG-BUILD-OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


@pytest.mark.parametrize("nlev", [72, 128])
def test_adaptive_cache_window_changes_the_policy_not_the_results(nlev):
    """include/caar_tuning.h "Adaptive window": the library times both cache policies of the default kernel on the host's own
    call pattern and keeps the faster.  The all-streaming policy launches the streaming twin of the default kernel (variant 1:
    another instantiation, POL = 1, its own XCD mapping) — results must not move: 130 calls with the adaptive window on (first
    probe at calls 48-61: seven all-streaming calls in between) equal 130 calls with the window forced, bit for bit, at both
    level counts that have a twin; afterwards the library reports a decision that agrees with what it measured, and forgets
    the set when its memory is released."""
    import ctypes as C
    import gc
    lib = tsa.library().lib
    E = 3000
    a = tsa.TestData().init_data(E, 4, nlev, device="cuda")
    b = tsa.TestData().init_data(E, 4, nlev, device="cuda")
    for d in (a, b):
        d.constants.eta_ave_w = 0.01   # (no rotation of the time levels: the np1 state is idempotent, the accumulators grow linearly)
    assert lib.caar_get_adaptive_window() == 1
    try:
        lib.caar_adaptive_window_reset()
        for _ in range(130):
            tsa.compute_and_apply_rhs(a)
        torch.cuda.synchronize()
        for _ in range(8):                # (the tuner looks at a finished probe's events at its next slot: within 8 calls)
            tsa.compute_and_apply_rhs(a)
        lib.caar_set_adaptive_window(0)
        for _ in range(138):
            tsa.compute_and_apply_rhs(b)
        torch.cuda.synchronize()
    finally:
        lib.caar_set_adaptive_window(1)
    for n in tsa.ARRAY_NAMES:
        assert torch.equal(a.arrays[n].view(torch.int64), b.arrays[n].view(torch.int64)), n
        assert torch.isfinite(a.arrays[n]).all(), n
    w, s, n = C.c_double(0), C.c_double(0), C.c_longlong(0)
    state = lib.caar_adaptive_window_state(C.c_void_p(a.arrays["elem_derived_vn0"].data_ptr()), C.byref(w), C.byref(s), C.byref(n))
    assert n.value >= 1 and w.value > 0 and s.value > 0, (state, w.value, s.value, n.value)
    assert state == (1 if w.value <= s.value * 1.003 else 0)
    # an array set no whole-range launch was seen of, and the forced-off switch
    assert lib.caar_adaptive_window_state(C.c_void_p(b.arrays["elem_derived_vn0"].data_ptr()), None, None, None) == -1
    # releasing the arrays drops the set's entry (a later allocation at the same address must not inherit its policy)
    key = a.arrays["elem_derived_vn0"].data_ptr()
    assert a.arrays.arena is not None
    del a
    gc.collect()
    assert lib.caar_adaptive_window_state(C.c_void_p(key), None, None, None) == -1
    lib.caar_adaptive_window_reset()
