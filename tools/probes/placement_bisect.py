"""Which array's allocation decides whether a data set is 'fast' or 'slow'?  Find a fast set F and a slow set S among
separately allocated ones, then time hybrids: S with one array taken from F, and F with one array taken from S."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, tinman_sandbox_amd as tsa
E, NP, NLEV = 10000, 4, 72
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
balg = tsa.algorithmic_bytes(NP, NLEV) * E
def rate(d, warm=40):
    def timed(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(n):
            tsa.compute_and_apply_rhs(d, st)
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    timed(warm)
    return balg / timed(20) / 8e7
def mix(base, other, names):
    t = dict(base.arrays.t)
    for n in names:
        t[n] = other.arrays.t[n]
    d = tsa.TestData().init_data(1, NP, NLEV, device=dev)
    d.arrays = tsa.ElementArrays(NP, NLEV, E, device=dev, tensors=t)
    d.control.nete = E
    return d
sets = [tsa.TestData().init_data(E, NP, NLEV, device=dev) for _ in range(4)]
rate(sets[0], 200)
rs = [rate(d) for d in sets]
while max(rs) < 1.02 * min(rs) and len(sets) < 40:
    sets.append(tsa.TestData().init_data(E, NP, NLEV, device=dev))
    rs.append(rate(sets[-1]))
print("sets: " + " ".join("%.1f" % r for r in rs))
F, S = sets[max(range(len(sets)), key=lambda i: rs[i])], sets[min(range(len(sets)), key=lambda i: rs[i])]
print("F %.1f  S %.1f" % (rate(F), rate(S)))
groups = {"state_v": ["elem_state_v"], "state_T": ["elem_state_T"], "state_dp3d": ["elem_state_dp3d"], "Qdp": ["elem_state_Qdp"],
          "vn0": ["elem_derived_vn0"], "omega_p": ["elem_derived_omega_p"], "phi": ["elem_derived_phi"], "pecnd": ["elem_derived_pecnd"],
          "eta": ["elem_derived_eta_dot_dpdn"],
          "geometry": ["elem_D", "elem_Dinv", "elem_fcor", "elem_spheremp", "elem_metdet", "elem_rmetdet", "elem_state_phis"]}
for g, names in groups.items():
    print("  %-10s S with F's: %.1f    F with S's: %.1f    (sizes MiB %s)" % (g, rate(mix(S, F, names)), rate(mix(F, S, names)),
          " ".join("%.0f" % (S.arrays.t[n].numel() * 8 / 2**20) for n in names)), flush=True)
allstate = ["elem_state_v", "elem_state_T", "elem_state_dp3d", "elem_state_Qdp"]
print("  all state  S with F's: %.1f    F with S's: %.1f" % (rate(mix(S, F, allstate)), rate(mix(F, S, allstate))))
for n in tsa.ARRAY_NAMES:
    print("   %-28s F 0x%x  S 0x%x" % (n, F.arrays.t[n].data_ptr(), S.arrays.t[n].data_ptr()))
