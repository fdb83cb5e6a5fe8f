#!/usr/bin/env python3
"""VALU instruction budget of one loop of a compiled kernel, by class.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -save-temps=obj -c tinman_sandbox_amd/csrc/caar_np4_steps.hip -o /tmp/x.o
    python tools/probes/isa_classes.py /tmp/caar_np4_steps-hip-amdgcn-amd-amdhsa-gfx950.s \\
        'caar_np4_steps_kernelILi72ELi5ELi2ELb1ELi0E' [--loop N] [--valu-per-element-call 2947.3] [--tiles 5 --tiles-per-element 18]

Takes the N-th (default: first) depth-1 loop of the kernel that contains inner loops — for the step-loop kernels that is the
steady call without stores (CARRY_IN = 11, STORES = 0) — and counts its instructions statically (every instruction once: the
body is straight-line tile code, five tile copies of which the fifth sits under a wave-uniform branch; the three inner loops
are the short sums over preceding tile totals).  Classes:
  fp64 arithmetic   v_fma/mul/add/...f64 minus the Newton steps of the reciprocals
  reciprocal        v_rcp_f64 + its 4 v_fma_f64 (recip(), caar_kernel_args.h)
  MFMA              v_mfma_f64_4x4x4 (issued on the matrix pipe; SQ_INSTS_MFMA)
  DPP moves         v_mov_b32_dpp (the in-tile scans: two per fp64 value and shift)
  selects / compares v_cndmask, v_cmp*
  moves             v_mov_b32 (non-DPP), v_accvgpr_*, v_readfirstlane / v_readlane
  integer / address everything else that starts with v_ (adds, shifts, mads on 32-bit values)
With --valu-per-element-call (SQ_INSTS_VALU per element-call from profiles/rNN/pmc_issue.json) the static shares are scaled
to the measured dynamic count; the flop minimum is SURVEY 8(d)'s 235 008 flop per element-call minus what the 180 MFMAs do
(92 160), as FMAs (2 flop per lane) and as single-flop operations."""
import argparse
import collections
import re

ap = argparse.ArgumentParser()
ap.add_argument("asm")
ap.add_argument("kernel")
ap.add_argument("--loop", type=int, default=0)
ap.add_argument("--valu-per-element-call", type=float, default=0.0)
ap.add_argument("--tiles", type=int, default=5)
ap.add_argument("--tiles-per-element", type=int, default=18)
a = ap.parse_args()

lines = open(a.asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and a.kernel in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end + 1]
# depth-1 loops: header label lines, extent = up to the last branch back to the header
heads = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l and l.startswith(".LBB")]
loops = []
for h in heads:
    label = body[h].split(":")[0]
    back = max(i for i, l in enumerate(body) if re.search(r"s_(c?branch\w*)\s+" + re.escape(label) + r"\b", l))
    inner = sum(1 for l in body[h:back] if "Inner Loop Header: Depth=2" in l)
    loops.append((h, back, inner, label))
cands = [x for x in loops if x[2] > 0] or loops
h, back, inner, label = cands[a.loop]
c = collections.Counter()
for l in body[h:back + 1]:
    l = l.strip()
    if not l or l.startswith((".", ";", "/")) or l.endswith(":"):
        continue
    op = l.split()[0]
    if op.startswith("v_mfma"):
        c["MFMA"] += 1
    elif op.startswith("v_rcp_f64"):
        c["rcp"] += 1
    elif op.startswith("v_") and "_f64" in op:
        c["f64:" + op.split("_e64")[0]] += 1
        c["f64"] += 1
    elif op.startswith("v_mov_b32") and ("dpp" in l or "row_" in l or "quad_perm" in l):
        c["dpp"] += 1
    elif op.startswith(("v_cndmask", "v_cmp")):
        c["select"] += 1
    elif op.startswith(("v_mov_b32", "v_accvgpr", "v_readfirstlane", "v_readlane", "v_writelane")):
        c["move"] += 1
    elif op.startswith("v_"):
        c["int"] += 1
        c["int:" + op.split("_e64")[0].split("_e32")[0]] += 1
    elif op.startswith("ds_bpermute"):
        c["ds_bpermute"] += 1
        c["LDS"] += 1
    elif op.startswith("ds_"):
        c["LDS"] += 1
    elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        c["VMEM"] += 1
    elif op.startswith("s_"):
        c["SALU"] += 1
newton = 4 * c["rcp"]
classes = [("fp64 arithmetic (fma / mul / add, Newton steps excluded)", c["f64"] - newton),
           ("reciprocal (v_rcp_f64 + 4 Newton fma each)", c["rcp"] + newton),
           ("DPP moves (v_mov_b32_dpp: the in-tile scans)", c["dpp"]),
           ("selects / compares (v_cndmask, v_cmp)", c["select"]),
           ("register moves (v_mov_b32, v_accvgpr_*, v_read*lane)", c["move"]),
           ("32-bit integer / address", c["int"])]
valu = sum(n for _, n in classes)
scale = (a.valu_per_element_call / valu) if a.valu_per_element_call else a.tiles_per_element / a.tiles
print("kernel %s\nloop %s: lines %d..%d of the kernel, %d inner loops" % (body[0].split(":")[0][:100], label, h, back, inner))
print("static: VALU %d (+ %d MFMA), LDS %d (of which ds_bpermute %d), VMEM %d, SALU %d" % (
    valu, c["MFMA"], c["LDS"], c["ds_bpermute"], c["VMEM"], c["SALU"]))
print("per element-call = static x %.3f (%s)" % (scale, "scaled to the measured SQ_INSTS_VALU" if a.valu_per_element_call
                                                else "tiles per element / tile copies in the body"))
print("| class | static (one wave, %d tiles) | per element-call | share |" % a.tiles)
print("|---|---|---|---|")
for name, n in classes:
    print("| %s | %d | %.0f | %.1f %% |" % (name, n, n * scale, 100.0 * n / valu))
print("| all VALU | %d | %.0f | 100 %% |" % (valu, valu * scale))
print("| v_mfma_f64_4x4x4 (matrix pipe) | %d | %.0f | |" % (c["MFMA"], c["MFMA"] * a.tiles_per_element / a.tiles))
flops_valu = 235008 - 180 * 512
print("arithmetic minimum of the VALU part: %d flop per element-call outside the MFMAs = %.0f VALU as FMAs, %.0f as single-flop ops"
      % (flops_valu, flops_valu / 128.0, flops_valu / 64.0))
print("fp64 opcodes:", {k[4:]: v for k, v in sorted(c.items()) if k.startswith("f64:")})
print("integer opcodes:", {k[4:]: v for k, v in sorted(c.items(), key=lambda x: -x[1]) if k.startswith("int:")})
