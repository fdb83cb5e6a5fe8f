"""CAAR data sets carved at consecutive positions of ONE huge allocation: is the rate a property of the physical region?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, tinman_sandbox_amd as tsa
E, NP, NLEV = 10000, 4, 72
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
shapes = tsa.array_shapes(NP, NLEV, 1, 3, E)
sizes = {n: int(torch.tensor(shapes[n]).prod()) for n in tsa.ARRAY_NAMES}
balg = tsa.algorithmic_bytes(NP, NLEV) * E
ref = tsa.TestData().init_data(E, NP, NLEV, device=dev)
SET = (sum(sizes.values()) * 8 + (32 << 20)) // (2 << 20) * (2 << 20) // 8   # doubles per position, 2 MiB multiple
NPOS = 20
big = torch.zeros(SET * NPOS, dtype=torch.float64, device=dev)
def carve(pos):
    off, tens = pos * SET + ((-(big.data_ptr() // 8)) % 32), {}
    for n in tsa.ARRAY_NAMES:
        off = (off + 31) // 32 * 32
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
print("torch's own allocations: %.1f" % rate(ref))
ds = [carve(p) for p in range(NPOS)]
for rnd in range(2):
    print("round %d, positions 0..%d of a %.0f GiB allocation: " % (rnd, NPOS - 1, SET * NPOS * 8 / 2**30) + " ".join("%.1f" % rate(d) for d in ds), flush=True)
