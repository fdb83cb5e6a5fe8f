"""Instruction mix of compiled kernels: python tools/probes/isa_mix.py <file.s> <mangled-name-substring> ...
(hipcc -save-temps=obj leaves the gfx950 assembly next to the object).  Counts per kernel body (straight-line
count, loops counted once): FP64 VALU by opcode, 32-bit VALU (DPP moves separately), LDS, VMEM, SALU."""
import collections
import sys

lines = open(sys.argv[1]).read().split("\n")


def body(sub):
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_ZN") and sub in l.split(":")[0] and ":" in l and not l.startswith(" "):
            start = i
            break
    if start is None:
        raise SystemExit("no kernel matching " + sub)
    out = []
    for l in lines[start + 1:]:
        out.append(l)
        if "s_endpgm" in l:
            break
    return lines[start].split(":")[0], out


for sub in sys.argv[2:]:
    name, b = body(sub)
    c = collections.Counter()
    for line in b:
        line = line.strip()
        if not line or line.startswith((".", ";", "/")) or line.endswith(":"):
            continue
        op = line.split()[0]
        dpp = "dpp" in line or "row_" in line or "quad_perm" in line
        if op.startswith("v_") and "_f64" in op:
            c["f64:" + op] += 1
            c["F64"] += 1
        elif op.startswith("v_"):
            c["VALU32"] += 1
            if dpp:
                c["dpp32"] += 1
        elif op.startswith("ds_"):
            c["LDS"] += 1
        elif op.startswith(("global_", "buffer_", "scratch_")):
            c["VMEM"] += 1
        elif op.startswith("s_"):
            c["SALU"] += 1
        else:
            c["other"] += 1
    print(name[:110])
    print("   ", {k: v for k, v in sorted(c.items()) if not k.startswith("f64:")})
    print("   ", {k[4:]: v for k, v in sorted(c.items()) if k.startswith("f64:")})
