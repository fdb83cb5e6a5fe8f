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

HIP_SOURCES = ["caar_np4.hip", "caar_np8.hip", "caar_abi.hip", "caar_norms.hip", "caar_membench.hip"]


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


def build_library(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [os.path.join(CSRC, "caar_kernel_args.h"),
                   os.path.join(HERE, "..", "include", "caar.h")]
    if force or _stale(LIB, deps):
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-shared",
               "-fno-gpu-rdc"] + srcs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB


def build_host_driver(force=False, verbose=False):
    """C++ host mirror of the reference driver (host/): links libcaar_hip.so."""
    srcs = [os.path.join(HOST, f) for f in sorted(os.listdir(HOST)) if f.endswith(".cpp")]
    if not srcs:
        return None
    exe = os.path.join(HOST, "caar_driver")
    deps = srcs + [os.path.join(HOST, f) for f in os.listdir(HOST) if f.endswith(".hpp")] + [LIB]
    if force or _stale(exe, deps):
        cmd = ["g++", "-std=c++17", "-O2", "-I" + os.path.join(HERE, "..", "include"), "-I" + HOST] + srcs + \
              ["-L" + CSRC, "-lcaar_hip", "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return exe


def build_all(force=False, verbose=False):
    lib = build_library(force, verbose)
    build_host_driver(force, verbose)
    return lib


if __name__ == "__main__":
    print(build_all(verbose=True))
