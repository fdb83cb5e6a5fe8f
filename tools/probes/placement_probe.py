"""Where does the allocation-to-allocation spread of the steady-state rate come from?
A: several 2.2 GB slabs alive at once, the 16 arrays carved out of each with the SAME internal layout.
B: one slab, different internal layouts (padding between consecutive arrays / 2 MiB alignment of every array)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, tinman_sandbox_amd as tsa
E, NP, NLEV = 10000, 4, 72
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
ref = tsa.TestData().init_data(E, NP, NLEV, device=dev)
shapes = tsa.array_shapes(NP, NLEV, 1, 3, E)
sizes = {n: int(torch.tensor(shapes[n]).prod()) for n in tsa.ARRAY_NAMES}
balg = tsa.algorithmic_bytes(NP, NLEV) * E
SLAB = (sum(sizes.values()) * 8 + (64 << 20) + 17 * (2 << 20)) // 8

def carve(big, pad_bytes=0, align=256):
    A = align // 8
    off = (-(big.data_ptr() // 8)) % A
    tens = {}
    for i, n in enumerate(tsa.ARRAY_NAMES):
        off = (off + A - 1) // A * A + (-(big.data_ptr() // 8) % A if False else 0)
        off += i * pad_bytes // 8 if pad_bytes else 0
        tens[n] = big[off: off + sizes[n]].view(shapes[n])
        tens[n].copy_(ref.arrays[n])
        off += sizes[n]
    d = tsa.TestData().init_data(1, NP, NLEV, device=dev)
    d.arrays = tsa.ElementArrays(NP, NLEV, E, device=dev, tensors=tens)
    d.control.nete = E
    return d

def rate(d):
    def timed(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(n):
            tsa.compute_and_apply_rhs(d, st)
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    timed(60)
    return balg / min(timed(20), timed(20)) / 8e7

print("torch's own 16 allocations (first in the process): %.1f %%" % rate(ref))
def tz(x):
    n = 0
    while x and not (x >> n) & 1:
        n += 1
    return n

import random
SLAB2 = SLAB + (16 * 40 << 20) // 8
slab = torch.zeros(SLAB2, dtype=torch.float64, device=dev)

def carve_gaps(big, gaps_mib):
    """array i starts at a 2 MiB-aligned offset, gaps_mib[i] extra MiB after the previous array's end"""
    A = (2 << 20) // 8
    off = (-(big.data_ptr() // 8)) % A
    tens, bases = {}, []
    for i, n in enumerate(tsa.ARRAY_NAMES):
        off = (off + A - 1) // A * A + gaps_mib[i] * (1 << 20) // 8
        tens[n] = big[off: off + sizes[n]].view(shapes[n])
        tens[n].copy_(ref.arrays[n])
        bases.append(off * 8)
        off += sizes[n]
    d = tsa.TestData().init_data(1, NP, NLEV, device=dev)
    d.arrays = tsa.ElementArrays(NP, NLEV, E, device=dev, tensors=tens)
    d.control.nete = E
    return d, bases

random.seed(7)
print("one slab, every array 2 MiB aligned, random extra gaps (multiples of 2 MiB, < 40 MiB) between arrays")
res = []
for trial in range(24):
    gaps = [0] * 16 if trial == 0 else [2 * random.randrange(0, 20) for _ in range(16)]
    d, bases = carve_gaps(slab, gaps)
    r = rate(d)
    res.append((r, gaps))
    print("  trial %2d: %.1f %%   gaps MiB %s" % (trial, r, gaps), flush=True)
