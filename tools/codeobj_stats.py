#!/usr/bin/env python3
"""What the shipped library holds, read from the library itself: every gfx950 kernel of libcaar_hip.so with its VGPR count,
spilled VGPRs, scratch and LDS bytes (the AMDGPU metadata notes of the code objects in the .hip_fatbin section).

    python tools/codeobj_stats.py [lib.so] [--all] [--json out.json]

Default output: kernel count per code object, every kernel that spills, and the BASELINE default kernels."""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    """[(triple, bytes)] of every device code object bundled into the file"""
    blob = open(path, "rb").read()
    out, pos = [], 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", blob, i + len(MAGIC))[0]
        p = i + len(MAGIC) + 8
        end = i + len(MAGIC)
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "amdgcn" in triple and size:
                out.append((triple, blob[i + off:i + off + size]))
            end = max(end, i + off + size)
        pos = max(end, i + len(MAGIC))
    return out


def kernels_of(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf_bytes)
        f.flush()
        notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
    ks = []
    for blk in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        g = lambda key: re.search(r"\.%s:\s+(\S+)" % key, blk)  # noqa: E731
        name = g("name").group(1)
        ks.append({"mangled": name, "vgprs": int(g("vgpr_count").group(1)), "vgpr_spills": int(g("vgpr_spill_count").group(1)),
                   "sgprs": int(g("sgpr_count").group(1)), "scratch_bytes": int(g("private_segment_fixed_size").group(1)),
                   "lds_bytes": int(g("group_segment_fixed_size").group(1))})
    if ks:
        dem = subprocess.run(["c++filt"] + [k["mangled"] for k in ks], capture_output=True, text=True).stdout.splitlines()
        for k, d in zip(ks, dem):
            k["name"] = d.replace("void caar::", "").split("(caar::")[0].split("(")[0] if "caar::" in d else d
    return ks


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else os.path.join(ROOT, "tinman_sandbox_amd", "csrc", "libcaar_hip.so")
    allk = []
    for triple, blob in code_objects(lib):
        ks = kernels_of(blob)
        caar = [k for k in ks if "caar_np" in k.get("name", "")]
        print("%s: code object of %d bytes, %d kernels (%d CAAR kernels)" % (triple, len(blob), len(ks), len(caar)))
        allk += ks
    caar = [k for k in allk if "caar_np" in k.get("name", "")]
    np4 = [k for k in caar if "caar_np4" in k["name"]]
    print("total: %d kernels, %d CAAR kernels (%d NP=4, %d NP=8)" % (len(allk), len(caar), len(np4), len(caar) - len(np4)))
    spill = [k for k in allk if k["vgpr_spills"] or k["scratch_bytes"]]
    print("kernels with register spills / scratch: %d" % len(spill))
    for k in spill:
        print("   %-100s vgprs %3d spilled %3d scratch %4d B  lds %6d B" % (k["name"][:100], k["vgprs"], k["vgpr_spills"], k["scratch_bytes"], k["lds_bytes"]))
    print("BASELINE default kernels:")
    for pat in ("caar_np4_kernel<72, 5, 1, true, 2, 0, false, false, false, 8, 0>", "caar_np4_kernel<128, 8, 2, true, 2, 0, false, false, false, 8, 27>",
                "caar_np8_kernel<72, 9, 1, true, true, false, false, false, false, true, 2>", "caar_np4_steps_kernel<72, 5, 2, true, 0,",
                "caar_np4_steps_kernel<128, 4, 2, true, 0,", "caar_np8_steps_kernel<72, 9, 1, true, false, 2>"):
        for k in allk:
            if k.get("name", "").startswith(pat):
                print("   %-100s vgprs %3d spilled %3d scratch %4d B  lds %6d B" % (k["name"][:100], k["vgprs"], k["vgpr_spills"], k["scratch_bytes"], k["lds_bytes"]))
    if "--all" in sys.argv:
        for k in allk:
            print("   %-110s vgprs %3d spilled %3d lds %6d" % (k.get("name", k["mangled"])[:110], k["vgprs"], k["vgpr_spills"], k["lds_bytes"]))
    if "--json" in sys.argv:
        json.dump(allk, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
