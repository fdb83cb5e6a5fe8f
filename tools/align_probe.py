#!/usr/bin/env python3
"""Does the relative placement of the 16 element arrays in HBM matter?  Carves them out of
one allocation with a configurable stagger between consecutive arrays and times the default
kernel (A/B in one process).  Layout INSIDE each array is untouched."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import tinman_sandbox_amd as tsa  # noqa: E402

E = 10000
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
ref = tsa.TestData().init_data(E, 4, 72, device=dev)
shapes = tsa.array_shapes(4, 72, 1, 3, E)


def carve(stagger_bytes, misalign=0, align=256):
    sizes = {n: int(torch.tensor(shapes[n]).prod()) for n in tsa.ARRAY_NAMES}
    total = sum(sizes.values()) + 136 * (stagger_bytes // 8) + 17 * (align // 8) + 4096
    big = torch.zeros(total, dtype=torch.float64, device=dev)
    A = align // 8
    base = (-(big.data_ptr() // 8)) % A  # align the start
    off = base
    tens = {}
    for i, n in enumerate(tsa.ARRAY_NAMES):
        off = base + (off - base + A - 1) // A * A   # every array `align`-byte aligned ...
        if misalign:
            off += misalign // 8             # ... or deliberately off by `misalign` bytes
        off += (i * stagger_bytes // 8)      # plus i * stagger between consecutive arrays
        tens[n] = big[off: off + sizes[n]].view(shapes[n])
        tens[n].copy_(ref.arrays[n])
        off += sizes[n]
    arr = tsa.ElementArrays(4, 72, E, device=dev, tensors=tens)
    d = tsa.TestData().init_data(1, 4, 72, device=dev)
    d.arrays = arr
    d.control.nete = E
    return d, big


def time_ms(d, reps=20):
    tsa.compute_and_apply_rhs(d, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        tsa.compute_and_apply_rhs(d, st)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


cands = [0, 4096 + 256]
data = {s: carve(s) for s in cands}
for m in (16, 64, 128):
    data["misaligned by %d B" % m] = carve(0, m)
for al in (4096, 65536, 2 * 1024 * 1024):
    data["aligned to %d" % al] = carve(0, 0, al)
data["torch"] = (ref, None)
for rnd in range(3):
    for s, (d, _) in data.items():
        ms = time_ms(d)
        print("round %d stagger %-20s %8.4f ms  %7.1f GB/s" % (rnd, s, ms, 213888 * E / ms / 1e6))
