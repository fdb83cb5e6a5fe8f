"""Workload for a PMC look at the placement effect: several placements of the same 10 000-element data set, 30 launches each,
in a fixed order (labels printed); run under rocprofv3 --pmc ... --kernel-trace and group the caar dispatches by 30."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, tinman_sandbox_amd as tsa
E, NP, NLEV = 10000, 4, 72
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
shapes = tsa.array_shapes(NP, NLEV, 1, 3, E)
sizes = {n: int(torch.tensor(shapes[n]).prod()) for n in tsa.ARRAY_NAMES}
SLAB = (sum(sizes.values()) * 8 + (64 << 20)) // 8
ref = tsa.TestData().init_data(E, NP, NLEV, device=dev)

def carve(big):
    A, off, tens = 32, (-(big.data_ptr() // 8)) % 32, {}
    for n in tsa.ARRAY_NAMES:
        off = (off + A - 1) // A * A
        tens[n] = big[off: off + sizes[n]].view(shapes[n])
        tens[n].copy_(ref.arrays[n])
        off += sizes[n]
    d = tsa.TestData().init_data(1, NP, NLEV, device=dev)
    d.arrays = tsa.ElementArrays(NP, NLEV, E, device=dev, tensors=tens)
    d.control.nete = E
    return d

def run(label, d):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        tsa.compute_and_apply_rhs(d, st)
    e0.record(st)
    for _ in range(20):
        tsa.compute_and_apply_rhs(d, st)
    e1.record(st)
    torch.cuda.synchronize()
    print("SET %s: %.4f ms" % (label, e0.elapsed_time(e1) / 20), flush=True)

run("torch-first", ref)
for gen in range(2):
    slabs = [torch.zeros(SLAB, dtype=torch.float64, device=dev) for _ in range(4)]
    for i, s in enumerate(slabs):
        run("gen%d-slab%d" % (gen, i), carve(s))
    del slabs
    torch.cuda.empty_cache()
more = [tsa.TestData().init_data(E, NP, NLEV, device=dev) for _ in range(3)]
for i, d in enumerate(more):
    run("torch-later%d" % i, d)
