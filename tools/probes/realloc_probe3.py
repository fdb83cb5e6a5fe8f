"""Placement effect: hybrid cache policy vs all-streaming on the same co-resident data sets (round robin)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, tinman_sandbox_amd as tsa
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
lib = tsa.library().lib
def timed(d, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        tsa.compute_and_apply_rhs(d, st)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
balg = tsa.algorithmic_bytes(4, 72) * 10000
sets = [tsa.TestData().init_data(10000, 4, 72, device=dev) for _ in range(6)]
timed(sets[0], 200)
for rnd in range(3):
    for var, name in ((0, "hybrid 192 MiB"), (1, "all streaming ")):
        lib.caar_select_variant(4, 72, var)
        out = []
        for d in sets:
            timed(d, 40)
            out.append(balg / timed(d, 20) / 8e7)
        print("round %d %s: %s" % (rnd, name, " ".join("%.1f" % x for x in out)), flush=True)
lib.caar_select_variant(4, 72, 0)
for n in ("elem_derived_vn0", "elem_derived_omega_p", "elem_derived_eta_dot_dpdn", "elem_state_v"):
    print(n, " ".join("0x%x" % d.arrays[n].data_ptr() for d in sets))
