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


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tinman_sandbox_amd", "csrc", "libcaar_hip.so")
    hs = kernel_hashes(lib)
    names = subprocess.run(["c++filt"] + list(hs), capture_output=True, text=True).stdout.splitlines()
    for mangled, dem in sorted(zip(hs, names), key=lambda x: x[1]):
        print("%s %6d  %s" % (hs[mangled][0], hs[mangled][1], dem.replace("void caar::", "").split("(caar::")[0]))
