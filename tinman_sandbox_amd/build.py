"""Builds the HIP extension in-tree for gfx950 (explicit hipcc, no JIT cache).

    python -m tinman_sandbox_amd.build            # libcaar_hip.so (+ the C++ host driver)

hipcc cross-compiles without a GPU; the built .so is git-ignored and travels
to the GPU box with the gpurun snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
LIB = os.path.join(CSRC, "libcaar_hip.so")
ARCH = "gfx950"

HIP_SOURCES = ["caar_np4.hip", "caar_np4_steps.hip", "caar_np8.hip", "caar_abi.hip", "caar_norms.hip", "caar_layout.hip", "caar_operators.hip", "caar_operators_ex.hip", "caar_alloc.hip", "caar_membench.hip"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X kernels cannot be built")
    return exe


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


LIB_DEBUG = os.path.join(CSRC, "libcaar_hip_debug.so")
# the translation units -DCAAR_DEBUG changes (the CAAR kernels and the ABI entry that reads their counters)
DEBUG_SOURCES = ("caar_np4.hip", "caar_np4_steps.hip", "caar_np8.hip", "caar_abi.hip")


LIB_EXTRA = os.path.join(CSRC, "libcaar_hip_extra.so")
# the translation units -DCAAR_EXTRA_NLEV=1 changes (csrc/caar_kernel_args.h: launch shapes specialised for seven more level
# counts, their step loops, the Eulerian form beyond 128 levels — not SURVEY section 8 rows, kept out of the default library)
EXTRA_SOURCES = ("caar_np4.hip", "caar_np4_steps.hip", "caar_abi.hip")


def build_library(force=False, verbose=False, jobs=4, debug=False, extra=False):
    """One object per .hip source (only stale ones are recompiled, `jobs` at a time), then one
    link: editing one kernel file costs one compile, not seven.  debug: libcaar_hip_debug.so, the same
    library with the kernels compiled with -DCAAR_DEBUG (the reference's check_dp3d as a device-side
    counter, include/caar.h caar_debug_dp3d_violations); the other objects are the release ones."""
    from concurrent.futures import ThreadPoolExecutor
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [
        os.path.join(HERE, "..", "include", "caar.h"), os.path.join(HERE, "..", "include", "caar_tuning.h")]
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"]
    LIB = LIB_DEBUG if debug else (LIB_EXTRA if extra else globals()["LIB"])

    def compile_one(name):
        dbg = debug and name in DEBUG_SOURCES
        ext = extra and name in EXTRA_SOURCES
        src, obj = os.path.join(CSRC, name), os.path.join(objdir, name.replace(".hip", ".debug.o" if dbg else (".extra.o" if ext else ".o")))
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc()] + flags + (["-DCAAR_DEBUG"] if dbg else []) + (["-DCAAR_EXTRA_NLEV=1"] if ext else []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(compile_one, HIP_SOURCES))
    if force or _stale(LIB, objs):
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


def build_host_driver(force=False, verbose=False, np_=4, nlev=72):
    """C++ host side (host/): libhomme_caar.so = Homme::compute_and_apply_rhs(TestData&) and
    the data model on top of libcaar_hip.so, and caar_driver, the reference driver's CLI."""
    inc = ["-I" + os.path.join(HERE, "..", "include"), "-I" + HOST]
    defs = ["-DCAAR_NP=%d" % np_, "-DCAAR_PLEV=%d" % nlev]
    suffix = "" if (np_, nlev) == (4, 72) else "_np%d_nlev%d" % (np_, nlev)
    shim = os.path.join(HOST, "libhomme_caar%s.so" % suffix)
    exe = os.path.join(HOST, "caar_driver%s" % suffix)
    hdrs = [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".hpp")] + [
        os.path.join(HERE, "..", "include", "caar.h"), os.path.join(HERE, "..", "include", "caar_tuning.h")]
    # $ORIGIN-relative run paths: the tree can be moved (the GPU box mounts it elsewhere)
    link = ["-L" + CSRC, "-lcaar_hip", "-Wl,-rpath,$ORIGIN/../csrc", "-Wl,-rpath,/opt/rocm/lib"]
    shim_srcs = [os.path.join(HOST, "homme_caar.cpp"), os.path.join(HOST, "homme_data.cpp")]
    if force or _stale(shim, shim_srcs + hdrs + [LIB]):
        cmd = ["g++", "-std=c++17", "-O2", "-fPIC", "-shared"] + defs + inc + shim_srcs + link + ["-o", shim]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    if force or _stale(exe, [os.path.join(HOST, "caar_driver.cpp"), shim] + hdrs):
        cmd = ["g++", "-std=c++17", "-O2"] + defs + inc + [os.path.join(HOST, "caar_driver.cpp")] + \
              ["-L" + HOST, "-lhomme_caar%s" % suffix, "-Wl,-rpath,$ORIGIN"] + link + ["-o", exe]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return exe


FLANG = "/opt/rocm/lib/llvm/bin/flang"


def build_fortran_driver(force=False, verbose=False):
    """Fortran host example (host/fortran/): caar_mod (iso_c_binding interface) + a driver with
    the reference Fortran driver's flow, linked against libcaar_hip.so.  Needs flang (ROCm)."""
    fdir = os.path.join(HOST, "fortran")
    if not os.path.exists(FLANG):
        return None
    out = os.path.join(fdir, "build")
    exe = os.path.join(out, "caar_f90_driver")
    srcs = [os.path.join(fdir, "caar_mod.F90"), os.path.join(fdir, "caar_f90_driver.F90")]
    if force or _stale(exe, srcs + [LIB]):
        os.makedirs(out, exist_ok=True)
        cmd = [FLANG, "-O2", "-module-dir", out] + srcs + ["-L" + CSRC, "-lcaar_hip",
                                                             "-Wl,-rpath,$ORIGIN/../../../csrc",
                                                             "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=out, capture_output=not verbose)
    return exe


def build_all(force=False, verbose=False, debug=True):
    lib = build_library(force, verbose)
    if debug:
        build_library(force, verbose, debug=True)
        build_library(force, verbose, extra=True)   # libcaar_hip_extra.so: the -DCAAR_EXTRA_NLEV=1 build, so that what it holds stays tested
    for np_, nlev in ((4, 72), (4, 128), (8, 72)):
        build_host_driver(force, verbose, np_, nlev)
    build_fortran_driver(force, verbose)
    return lib


if __name__ == "__main__":
    import sys
    if "--debug" in sys.argv[1:]:
        print(build_library(verbose=True, debug=True))
    else:
        print(build_all(verbose=True))
