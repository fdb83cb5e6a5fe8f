#!/usr/bin/env python3
"""Fingerprint of every gfx950 kernel a library ships: md5 of its disassembled instruction stream (addresses and
encodings stripped, so a kernel that merely moved inside the code object keeps its hash).

    python tools/isa_hash.py [lib.so] > before.txt ; ...edit, rebuild... ; python tools/isa_hash.py > after.txt ; diff

Used when source is refactored without intent to change code (round 5: removal of the compile-time experiment paths)."""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from codeobj_stats import ROOT, code_objects  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def kernel_hashes(path):
    out = {}
    for _, elf in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
        name, h, n = None, None, 0
        for ln in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
            if m:
                if name:
                    out[name] = (h.hexdigest(), n)
                name, h, n = m.group(1), hashlib.md5(), 0
                continue
            if name and ln.strip():
                ins = re.sub(r"//.*$", "", ln).strip()            # drop the address comment
                ins = re.sub(r"<[^>]*>", "", ins)                  # and symbolic branch targets (kept: the numeric offset)
                h.update(ins.encode())
                n += 1
        if name:
            out[name] = (h.hexdigest(), n)
    return out


def demangled_hashes(lib):
    """{demangled kernel name (as the variant tables spell it): (md5, instruction count)}"""
    hs = kernel_hashes(lib)
    names = subprocess.run(["c++filt"] + list(hs), capture_output=True, text=True).stdout.splitlines()
    return {dem.replace("void caar::", "").split("(caar::")[0]: hs[m] for m, dem in zip(hs, names)}


def compiler_id():
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout
    return " | ".join(ln.strip() for ln in out.splitlines()[:2])


# the kernels a default launch of the BASELINE configurations reaches (single calls, all-streaming twins, step loops)
GUARDED = ("caar_np4_kernel<72, 5, 1, true, 2, 0, false, false, false, 8, 0>",
           "caar_np4_kernel<72, 5, 1, true, 1, 0, false, false, false, 8, 0>",
           "caar_np4_kernel<128, 8, 2, true, 2, 0, false, false, false, 8, 27>",
           "caar_np4_kernel<128, 8, 2, true, 1, 0, false, false, false, 8, 27>",
           "caar_np8_kernel<72, 9, 1, true, true, false, false, false, false, true, 2>",
           "caar_np4_steps_kernel<72, 5, 2, true, 0, 0, 0, 4, 0>",
           "caar_np4_steps_kernel<128, 4, 2, true, 0, 0, 0, 3, 0>",
           "caar_np8_steps_kernel<72, 9, 1, true, false, 2>")


if __name__ == "__main__":
    args = [x for x in sys.argv[1:] if not x.startswith("--")]
    lib = args[0] if args else os.path.join(ROOT, "tinman_sandbox_amd", "csrc", "libcaar_hip.so")
    hs = demangled_hashes(lib)
    if "--write-golden" in sys.argv:   # tests/golden/isa_fingerprints.json: what tests/test_host.py holds the build against
        import json
        out = {"compiler": compiler_id(), "kernels": {k: {"md5": hs[k][0], "instructions": hs[k][1]} for k in GUARDED}}
        json.dump(out, open(os.path.join(ROOT, "tests", "golden", "isa_fingerprints.json"), "w"), indent=1)
        print("wrote tests/golden/isa_fingerprints.json for", out["compiler"])
    else:
        for dem in sorted(hs):
            print("%s %6d  %s" % (hs[dem][0], hs[dem][1], dem))
