"""Inside ONE huge allocation: the 16 arrays of a data set placed `spacing` apart (array i at i * spacing).  Tests whether the
rate depends on high address bits of the arrays relative to each other."""
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
GB = int(sys.argv[1]) if len(sys.argv) > 1 else 80
big = torch.zeros(GB << 27, dtype=torch.float64, device=dev)
print("allocation of %d GiB at 0x%x" % (GB, big.data_ptr()))
def carve(offsets_bytes):
    tens = {}
    for n, ob in zip(tsa.ARRAY_NAMES, offsets_bytes):
        off = ob // 8
        tens[n] = big[off: off + sizes[n]].view(shapes[n])
        tens[n].copy_(ref.arrays[n])
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
MiB, GiB = 1 << 20, 1 << 30
def packed(start):
    offs, o = [], start
    for n in tsa.ARRAY_NAMES:
        offs.append(o)
        o += (sizes[n] * 8 + 2 * MiB - 1) // (2 * MiB) * (2 * MiB)
    return offs
print("packed: %.1f" % rate(carve(packed(0))), flush=True)
def two_clusters(a_gib, b_gib, which_b):
    """arrays listed in which_b packed from b_gib, the others packed from a_gib"""
    offs, oa, ob = [], int(a_gib * GiB), int(b_gib * GiB)
    for n in tsa.ARRAY_NAMES:
        sz = (sizes[n] * 8 + 2 * MiB - 1) // (2 * MiB) * (2 * MiB)
        if n in which_b:
            offs.append(ob); ob += sz
        else:
            offs.append(oa); oa += sz
    return offs
names = list(tsa.ARRAY_NAMES)
alt = names[1::2]
half = ["elem_state_v", "elem_state_T", "elem_derived_vn0", "elem_derived_phi", "elem_derived_eta_dot_dpdn"]
for a, b in ((0, 16), (0, 28), (0, 33), (0, 40), (0, 64), (8, 24), (20, 44), (33, 49), (34, 70)):
    print("two compact clusters at %2d and %2d GiB: alternate arrays %.1f   5 big arrays %.1f" % (
        a, b, rate(carve(two_clusters(a, b, alt))), rate(carve(two_clusters(a, b, half)))), flush=True)
for start in (30.0, 31.0, 31.5, 32.0, 33.0, 62.0, 63.0, 63.5):
    print("packed from %.1f GiB: %.1f" % (start, rate(carve(packed(int(start * GiB))))), flush=True)
