"""CPU-side tests of the host logic and of the C-ABI library's surface (no GPU:
nothing here launches a kernel)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import pyoracle as po

import tinman_sandbox_amd as tsa
from tinman_sandbox_amd import build as tbuild
from tinman_sandbox_amd import caar as m

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    tbuild.build_library()
    return tsa.library()


def _declared(header):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(caar_[a-z_0-9]+)\s*\(", hdr))


def test_library_exports_every_declared_symbol(lib):
    """Every function include/caar.h (the frozen boundary) and include/caar_tuning.h declare is exported by libcaar_hip.so,
    the two headers are disjoint, and the Python binding lists exactly the same two sets."""
    boundary, tuning = _declared("caar.h"), _declared("caar_tuning.h")
    assert boundary and tuning, "no declarations parsed"
    assert not (boundary & tuning)
    assert boundary == set(m.CaarLibrary.BOUNDARY_SYMBOLS)
    assert tuning == set(m.CaarLibrary.TUNING_SYMBOLS)
    for s in boundary | tuning:
        assert hasattr(lib.lib, s), s
    import __graft_entry__ as entry
    assert lib.lib.caar_abi_version() == entry.header_abi_version() == 6


def test_boundary_is_what_integration_md_lists_and_what_the_hosts_bind():
    """include/caar.h is frozen at ABI 6.  Its exported-symbol set equals the list in INTEGRATION.md
    section 3 ("The frozen boundary"), and the host bindings use nothing outside it: the undefined caar_* symbols of
    host/libhomme_caar.so (the C++ shim the reference's main.cpp links) and the bind(C) names of host/fortran/caar_mod.F90
    are subsets of it."""
    import subprocess
    boundary = _declared("caar.h")
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("### The frozen boundary"):]
    sec = sec[:sec.index("\n## ")] if "\n## " in sec else sec
    listed = set(re.findall(r"`(caar_[a-z_0-9]+)`", sec[:sec.index("### Tuning")]))
    assert listed == boundary, (sorted(listed - boundary), sorted(boundary - listed))
    tbuild.build_host_driver()
    shim = os.path.join(ROOT, "tinman_sandbox_amd", "host", "libhomme_caar.so")
    nm = subprocess.run(["nm", "-D", "--undefined-only", shim], check=True, capture_output=True, text=True).stdout
    used = set(re.findall(r"\bU (caar_[a-z_0-9]+)", nm))
    assert used and used <= boundary, sorted(used - boundary)
    f90 = open(os.path.join(ROOT, "tinman_sandbox_amd", "host", "fortran", "caar_mod.F90")).read()
    bound = set(re.findall(r'bind\(C,\s*name="(caar_[a-z_0-9]+)"\)', f90))
    assert bound and bound <= boundary, sorted(bound - boundary)
    # the shim's sources name only the boundary header
    for f in ("homme_caar.cpp", "homme_caar.hpp"):
        assert "caar_tuning.h" not in open(os.path.join(ROOT, "tinman_sandbox_amd", "host", f)).read(), f


def test_graft_entry_build_succeeds_on_a_built_tree(capsys):
    """The driver's official "does it build" entry point: compiles whatever is stale (nothing,
    on a built tree), loads the library and checks it against include/caar.h."""
    import __graft_entry__ as entry
    entry.build()
    assert "built" in capsys.readouterr().out


def test_supported_variants_and_names(lib):
    assert lib.lib.caar_supported(4, 72) == 1
    assert lib.lib.caar_supported(4, 128) == 1
    assert lib.lib.caar_supported(5, 72) == 0
    for nlev in (26, 30, 32, 60, 64, 80, 96):
        assert lib.lib.caar_supported(4, nlev) == 1
    assert lib.lib.caar_supported(4, 27) == 1 and lib.lib.caar_supported(4, 256) == 1   # run-time level count
    assert lib.lib.caar_supported(4, 1) == 0 and lib.lib.caar_supported(4, 257) == 0
    assert lib.lib.caar_supported(8, 128) == 0
    # per vertical form (ABI 6): the default library has no Eulerian form beyond 128 levels
    for np_, nlev in ((4, 72), (4, 128), (4, 27), (8, 72)):
        assert lib.lib.caar_supported_ex(np_, nlev, 1) == 1 and lib.lib.caar_supported_ex(np_, nlev, 0) == 1
    assert lib.lib.caar_supported_ex(4, 129, 1) == 1 and lib.lib.caar_supported_ex(4, 256, 1) == 1
    assert lib.lib.caar_supported_ex(4, 129, 0) == 0 and lib.lib.caar_supported_ex(4, 256, 0) == 0
    assert lib.lib.caar_supported_ex(5, 72, 1) == 0 and lib.lib.caar_supported_ex(4, 72, -1) == 0
    extra = os.path.join(ROOT, "tinman_sandbox_amd", "csrc", "libcaar_hip_extra.so")
    if os.path.exists(extra):
        X = C.CDLL(extra)
        assert X.caar_supported_ex(4, 200, 0) == 1 and X.caar_abi_version() == 6
    assert b"caar" in lib.lib.caar_kernel_name(4, 72)
    assert lib.lib.caar_kernel_name(3, 3) is None
    assert lib.lib.caar_strerror(-2).decode().startswith("no kernel")


@pytest.mark.parametrize("np_,nlev,ne", [(4, 72, 3), (4, 128, 2), (8, 72, 5)])
def test_array_lengths_and_algorithmic_bytes(lib, np_, nlev, ne):
    dims = m._CaarDims(np_, nlev, 1, 3, ne)
    shapes = tsa.array_shapes(np_, nlev, 1, 3, ne)
    for i, n in enumerate(tsa.ARRAY_NAMES):
        assert lib.lib.caar_array_len(C.byref(dims), i) == int(np.prod(shapes[n])), n
    assert lib.lib.caar_array_len(C.byref(dims), 16) == -1
    assert lib.lib.caar_algorithmic_bytes(np_, nlev, 0) == tsa.algorithmic_bytes(np_, nlev)
    assert lib.lib.caar_algorithmic_bytes(np_, nlev, 1) == tsa.algorithmic_bytes(np_, nlev, dry=True)
    # SURVEY.md 8d table
    assert tsa.algorithmic_bytes(4, 72) == 213888
    assert tsa.algorithmic_bytes(4, 128) == 378752
    assert tsa.algorithmic_bytes(8, 72) == 855552


def test_launch_validates_without_touching_a_device(lib):
    """Argument errors are detected before any HIP call (safe on a CPU-only box)."""
    dims = m._CaarDims(4, 72, 1, 3, 4)
    prm = m._CaarParams(0, 5, 0, 1, 2, 0, 1.0, 1.0, 1.0, 461.5, 287.04, 0.28, 10.0, 73.0, None)
    ptrs = m._CaarArrays()
    assert lib.lib.caar_launch(C.byref(dims), C.byref(ptrs), None, C.byref(prm), None) == -1  # nete > num_elems
    prm.nete = 4
    assert lib.lib.caar_launch(C.byref(dims), C.byref(ptrs), None, C.byref(prm), None) == -1  # null arrays
    dims2 = m._CaarDims(6, 72, 1, 3, 4)
    fake = m._CaarArrays(*[C.cast(64, m._dp)] * 16)
    assert lib.lib.caar_launch(C.byref(dims2), C.byref(fake), C.c_void_p(64), C.byref(prm), None) == -2
    assert lib.lib.caar_launch(None, None, None, None, None) == -1
    # rsplit == 0 beyond 128 levels: refused up front with CAAR_EUNSUPPORTED (no launch is attempted)
    dims3 = m._CaarDims(4, 200, 1, 3, 4)
    prm3 = m._CaarParams(0, 4, 0, 1, 2, 0, 1.0, 1.0, 1.0, 461.5, 287.04, 0.28, 10.0, 201.0, None, 0, None, C.cast(64, m._dp))
    assert lib.lib.caar_launch(C.byref(dims3), C.byref(fake), C.c_void_p(64), C.byref(prm3), None) == -2
    assert lib.lib.caar_launch_steps(C.byref(dims3), C.byref(fake), C.c_void_p(64), C.byref(prm3), 3, 1, None) == -2


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(m.CaarError, match="no CPU fallback"):
        m.CaarLibrary(str(tmp_path / "nope.so"))


def test_cpu_arrays_are_refused():
    d = tsa.TestData().init_data(2, 4, 72, device="cpu")
    with pytest.raises(m.CaarError, match="no CPU fallback"):
        tsa.compute_and_apply_rhs(d)


@pytest.mark.parametrize("np_,nlev", [(4, 72), (4, 128), (8, 72)])
def test_closed_form_init_matches_oracle_init(oracle, np_, nlev):
    """ElementArrays.init_data == Arrays::init_data (data_structures.cpp:42-92)."""
    ne = 3
    mine = tsa.ElementArrays(np_, nlev, ne, device="cpu").init_data().to_numpy()
    ref = oracle.init_arrays(np_, nlev, 1, 3, ne)
    for n in tsa.ARRAY_NAMES:
        assert np.array_equal(mine[n], ref[n]), n
    # a rank's slab [first, first+n) is the matching slice of the global arrays
    slab = tsa.ElementArrays(np_, nlev, 2, device="cpu").init_data(first_elem=1).to_numpy()
    for n in tsa.ARRAY_NAMES:
        assert np.array_equal(slab[n], ref[n][1:3]), n


def test_reference_scalars_and_dvv(oracle):
    d = tsa.TestData().init_data(2, 4, 72, device="cpu")
    sc = po.default_scalars(72)
    assert (d.control.n0, d.control.np1, d.control.nm1, d.control.qn0) == (0, 1, 2, 0)
    for k in ("rrearth", "eta_ave_w", "Rwater_vapor", "Rgas", "kappa"):
        assert getattr(d.constants, k) == sc[k]
    assert d.hvcoord.ps0 == sc["ps0"] and np.array_equal(d.hvcoord.hyai, sc["hyai"])
    assert np.array_equal(d.deriv.Dvv, oracle.dvv_np4(False))
    assert np.array_equal(tsa.Derivative(4, f32_rounded=True).Dvv, oracle.dvv_np4(True))
    assert np.max(np.abs(tsa.gll_derivative_matrix(4) - oracle.dvv_np4(False))) < 2e-15
    assert np.max(np.abs(tsa.gll_derivative_matrix(8) - oracle.dvv_gll(8))) < 5e-14
    d.update_time_levels()  # data_structures.cpp:174-180
    assert (d.control.n0, d.control.np1, d.control.nm1) == (1, 2, 0)


def test_shard_range_partitions_exactly():
    for E, G in [(10000, 1), (100000, 8), (10, 3), (5, 8), (64, 2)]:
        seen = []
        for r in range(G):
            a, b = tsa.shard_range(E, r, G)
            assert 0 <= a <= b <= E
            seen += list(range(a, b))
        assert seen == list(range(E))
    assert tsa.shard_range(100000, 3, 8) == (37500, 50000)


def test_header_is_plain_c_and_links_from_c(tmp_path, lib):
    """include/caar.h and include/caar_tuning.h must be usable from C (the ABI is a C ABI): compile a C99 translation
    unit against each with gcc -Wall -Werror (caar.h on its own first: the boundary must not need the tuning header),
    link to libcaar_hip.so and run the calls that need no GPU."""
    import subprocess
    src = tmp_path / "abi_probe.c"
    src.write_text(r'''
#include <stdio.h>
#include "caar.h"
#ifdef WITH_TUNING
#include "caar_tuning.h"
#endif
int main(void) {
  CaarDims d = {4, 72, 1, 3, 10};
  CaarParams p = {0};
  CaarArrays a = {0};
  if (caar_abi_version() != CAAR_ABI_VERSION) return 1;
  if (!caar_supported(4, 72) || caar_supported(3, 3)) return 2;
  if (caar_array_len(&d, 7) != 10LL * 3 * 72 * 16 * 2) return 3;
  if (caar_algorithmic_bytes(4, 72, 0) != 213888) return 4;
  p.nete = 11; /* > num_elems: refused before any device is touched */
  if (caar_launch(&d, &a, 0, &p, 0) != CAAR_EINVAL) return 5;
  if (!caar_supported_ex(4, 128, 0) || caar_supported_ex(4, 129, 0)) return 6;
#ifdef WITH_TUNING
  if (caar_get_cache_window() != CAAR_CACHE_WINDOW_DEFAULT) return 7;
  printf("%s|", caar_kernel_name(4, 72));
#endif
  printf("%s\n", caar_strerror(CAAR_EUNSUPPORTED));
  return 0;
}
''')
    exe = tmp_path / "abi_probe"
    csrc = os.path.join(ROOT, "tinman_sandbox_amd", "csrc")
    for defs in ([], ["-DWITH_TUNING"]):
        subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include")] + defs +
                       [str(src), "-L" + csrc, "-lcaar_hip", "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib",
                        "-o", str(exe)], check=True)
        out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
        assert "no kernel" in out and (("caar_np4_kernel<72" in out) == bool(defs))


def test_shipped_code_object_holds_what_design_says():
    """DESIGN.md section 7 (round 4): the default library holds the BASELINE configurations and the run-time-level kernel —
    fewer than 100 CAAR kernel instantiations — and every kernel is free of register spills except the documented one
    (the Eulerian NLEV=128 default).  Read from the library itself (tools/codeobj_stats.py: AMDGPU metadata notes)."""
    import importlib.util
    import shutil
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf") or shutil.which("c++filt") is None:
        pytest.skip("llvm-readelf / c++filt not available")
    spec = importlib.util.spec_from_file_location("codeobj_stats", os.path.join(ROOT, "tools", "codeobj_stats.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    lib = os.path.join(ROOT, "tinman_sandbox_amd", "csrc", "libcaar_hip.so")
    kernels = [k for _, blob in m.code_objects(lib) for k in m.kernels_of(blob)]
    caar = [k for k in kernels if "caar_np" in k.get("name", "")]
    np4 = [k for k in caar if "caar_np4" in k["name"]]
    assert 40 <= len(np4) < 100, len(np4)
    spilling = sorted(k["name"] for k in kernels if k["vgpr_spills"] or k["scratch_bytes"])
    assert spilling == ["caar_np4_kernel<128, 8, 2, true, 2, 0, false, false, true, 8, 33>"], spilling
    # the default kernels of the BASELINE configurations and their step loops
    by_name = {k["name"]: k for k in caar}
    for name, lds_max in (("caar_np4_kernel<72, 5, 1, true, 2, 0, false, false, false, 8, 0>", 80 << 10),
                          ("caar_np4_kernel<128, 8, 2, true, 2, 0, false, false, false, 8, 27>", 80 << 10),
                          ("caar_np8_kernel<72, 9, 1, true, true, false, false, false, false, true, 2>", 160 << 10)):
        k = by_name[name]
        assert k["vgprs"] <= 256 and k["vgpr_spills"] == 0 and k["lds_bytes"] <= lds_max, k
    assert any(n.startswith("caar_np4_steps_kernel<72,") for n in by_name) and any(n.startswith("caar_np8_steps_kernel<72,") for n in by_name)


def test_window_tuner_state_machine_under_a_fake_clock(tmp_path):
    """csrc/caar_window_tuner.h is free of HIP: tests/window_tuner_sim.cpp drives the adaptive cache window's state machine
    with a simulated device (replaying host, evicting neighbour, a pattern change, a discarded probe, a host that enqueues
    300 launches ahead, ties) and checks the decisions, when they are made, and how rarely a launch needs the slow path."""
    import subprocess
    exe = str(tmp_path / "window_tuner_sim")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "tinman_sandbox_amd", "csrc"),
                    os.path.join(ROOT, "tests", "window_tuner_sim.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_shipped_headline_kernels_are_the_measured_instruction_streams():
    """the kernels' speed is pinned to what one hipcc makes of a 17-parameter template.  The GPU guard
    (tests/test_bench_gpu.py::test_performance_lower_bounds) catches a slow build on the GPU; this one catches a CHANGED build on
    the CPU: the instruction streams of the kernels a default launch of the BASELINE configurations reaches (tools/isa_hash.py:
    md5 over the disassembly, addresses stripped) must be the ones the committed measurements were taken with
    (tests/golden/isa_fingerprints.json).  A deliberate change of a kernel re-measures (tools/perf_guard.py, bench.py) and then
    refreshes the file with `python tools/isa_hash.py --write-golden`.  Another compiler: skipped, the GPU guard decides."""
    import importlib.util
    import json
    import shutil
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump") or shutil.which("c++filt") is None:
        pytest.skip("llvm-objdump / c++filt not available")
    tools = os.path.join(ROOT, "tools")
    sys_path_added = tools not in __import__("sys").path
    if sys_path_added:
        __import__("sys").path.insert(0, tools)
    spec = importlib.util.spec_from_file_location("isa_hash", os.path.join(tools, "isa_hash.py"))
    ih = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ih)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "isa_fingerprints.json")))
    if ih.compiler_id() != gold["compiler"]:
        pytest.skip("built by another compiler than the fingerprints were taken with: %s" % ih.compiler_id())
    tbuild.build_library()
    got = ih.demangled_hashes(os.path.join(ROOT, "tinman_sandbox_amd", "csrc", "libcaar_hip.so"))
    assert set(ih.GUARDED) == set(gold["kernels"])
    changed = [k for k, v in gold["kernels"].items() if k not in got or got[k][0] != v["md5"]]
    assert not changed, "instruction streams changed (re-measure, then `python tools/isa_hash.py --write-golden`): %s" % changed
