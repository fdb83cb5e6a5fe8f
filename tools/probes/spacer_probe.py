"""Fresh process: the 16 arrays allocated separately, in two groups (alternate arrays), with a temporary spacer allocation of S GiB
between the groups (freed afterwards).  Does the data set reach the fast level?   python spacer_probe.py S [ngroups]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, tinman_sandbox_amd as tsa
S = float(sys.argv[1])
NG = int(sys.argv[2]) if len(sys.argv) > 2 else 2
E, NP, NLEV = 10000, 4, 72
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
shapes = tsa.array_shapes(NP, NLEV, 1, 3, E)
balg = tsa.algorithmic_bytes(NP, NLEV) * E
names = list(tsa.ARRAY_NAMES)
CHURN = len(sys.argv) > 3 and sys.argv[3] == "churn"
if CHURN:  # what a process looks like after other work: allocate, use and free a few large data sets first
    for np_, nlev, e in ((4, 72, 10000), (8, 72, 20000), (4, 128, 12500)):
        t = tsa.TestData().init_data(e, np_, nlev, device=dev)
        tsa.compute_and_apply_rhs(t, st)
        torch.cuda.synchronize()
        del t
        torch.cuda.empty_cache()
tens, spacers = {}, []
for g in range(NG):
    for i, n in enumerate(names):
        if i % NG == g:
            tens[n] = torch.zeros(shapes[n], dtype=torch.float64, device=dev)
    if S > 0 and g < NG - 1:
        spacers.append(torch.empty(int(S * (1 << 30)), dtype=torch.uint8, device=dev))
del spacers
torch.cuda.empty_cache()
d = tsa.TestData().init_data(1, NP, NLEV, device=dev)
arr = tsa.ElementArrays(NP, NLEV, E, device=dev, tensors=tens)
arr.init_data(0)
d.arrays = arr
d.control.nete = E
def timed(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        tsa.compute_and_apply_rhs(d, st)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
timed(150)
r = [balg / timed(20) / 8e7 for _ in range(3)]
print("%sspacer %5.1f GiB x %d groups: %s %% of peak" % ("after churn, " if CHURN else "", S, NG, " ".join("%.1f" % x for x in r)), flush=True)
