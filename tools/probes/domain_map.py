"""Map the 'domains' of device memory as the CAAR kernel sees them: one huge allocation; the data set split into two compact
clusters (alternate arrays), one fixed at offset 0, the other at offset r: rate(r) is low when r lies in the same domain as offset 0."""
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
GB = int(sys.argv[1]) if len(sys.argv) > 1 else 240
big = torch.empty(GB << 27, dtype=torch.float64, device=dev)
print("allocation of %d GiB at 0x%x" % (GB, big.data_ptr()), flush=True)
MiB, GiB = 1 << 20, 1 << 30
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
    timed(40)
    return balg / min(timed(20), timed(20)) / 8e7
names = list(tsa.ARRAY_NAMES)
def clusters(starts_gib, assign):
    """assign[n] -> cluster index; each cluster packed from its start"""
    o = [int(s * GiB) for s in starts_gib]
    offs = []
    for n in names:
        c = assign[n]
        offs.append(o[c])
        o[c] += (sizes[n] * 8 + 2 * MiB - 1) // (2 * MiB) * (2 * MiB)
    return offs
alt2 = {n: i % 2 for i, n in enumerate(names)}
A = 0
cand = list(range(16, GB - 4, 16))
base = rate(carve(clusters([0, 4], alt2)))
rel0 = {r: rate(carve(clusters([0, r], alt2))) for r in cand}
print("packed-ish (0, 4): %.1f; relative to 0: " % base + " ".join("%d:%.1f" % kv for kv in rel0.items()), flush=True)
B = next(r for r in cand if rel0[r] > base + 1.5)
relB = {r: rate(carve(clusters([B, r if r != B else B + 4], alt2))) for r in cand}
print("relative to %d: " % B + " ".join("%d:%.1f" % kv for kv in relB.items()), flush=True)
third = [r for r in cand if rel0[r] > base + 1.5 and relB[r] > base + 1.5]
print("B = %d GiB; regions different from both 0 and B: %s" % (B, third), flush=True)
import random
random.seed(3)
big9 = ["elem_state_dp3d", "elem_state_v", "elem_state_T", "elem_state_Qdp", "elem_derived_eta_dot_dpdn", "elem_derived_omega_p",
        "elem_derived_phi", "elem_derived_pecnd", "elem_derived_vn0"]
short = {"elem_state_dp3d": "dp", "elem_state_v": "v", "elem_state_T": "T", "elem_state_Qdp": "Q", "elem_derived_eta_dot_dpdn": "eta",
         "elem_derived_omega_p": "om", "elem_derived_phi": "phi", "elem_derived_pecnd": "pec", "elem_derived_vn0": "vn0"}
res = []
def run(assign9, tag=""):
    assign = {n: 0 for n in names}
    assign.update(assign9)
    r = rate(carve(clusters([A, B], assign)))
    res.append((r, " ".join(short[n] for n in big9 if assign9[n] == 1), tag))
run({n: 0 for n in big9}, "all in one domain")
run({n: (1 if n in ("elem_state_T", "elem_derived_vn0", "elem_derived_phi") else 0) for n in big9}, "T vn0 phi")
run({n: (1 if n in ("elem_state_T", "elem_derived_vn0", "elem_derived_phi", "elem_state_v") else 0) for n in big9}, "T vn0 phi v")
run({n: (1 if n in ("elem_state_v",) else 0) for n in big9}, "only v")
run({n: (1 if n in ("elem_state_dp3d", "elem_state_v", "elem_state_T") else 0) for n in big9}, "state vs derived")
run({n: (1 if n in ("elem_state_Qdp", "elem_derived_pecnd") else 0) for n in big9}, "read-only arrays apart")
for _ in range(40):
    run({n: random.randrange(2) for n in big9})
if third:
    Cc = third[0]
    three = {n: i % 3 for i, n in enumerate(names)}
    print("three clusters (array i -> i mod 3) at 0, %d, %d GiB: %.1f" % (B, Cc, rate(carve(clusters([0, B, Cc], three)))), flush=True)
    eq = {"elem_state_dp3d": 0, "elem_state_T": 1, "elem_state_v": 2, "elem_state_Qdp": 0, "elem_derived_vn0": 1, "elem_derived_omega_p": 0,
          "elem_derived_phi": 1, "elem_derived_pecnd": 2, "elem_derived_eta_dot_dpdn": 2}
    a3 = {n: eq.get(n, i % 3) for i, n in enumerate(names)}
    print("three clusters, equal-stride arrays apart: %.1f" % rate(carve(clusters([0, B, Cc], a3))), flush=True)
    if len(third) > 1:
        four = {n: i % 4 for i, n in enumerate(names)}
        print("four clusters at 0, %d, %d, %d: %.1f" % (B, Cc, third[-1], rate(carve(clusters([0, B, Cc, third[-1]], four)))), flush=True)
res.sort()
for r, s_, tag in res:
    print("  %.1f   in the other domain: %-40s %s" % (r, s_, tag))
sys.exit(0)

rate(carve(clusters([0, 2], alt2)))
print("second cluster at r GiB (first at 0): rate")
line = []
for r in list(range(2, GB - 2, 4)):
    line.append("%d:%.1f" % (r, rate(carve(clusters([0, r], alt2)))))
    if len(line) == 12:
        print("  " + "  ".join(line), flush=True)
        line = []
print("  " + "  ".join(line), flush=True)
