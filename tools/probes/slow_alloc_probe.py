import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, tinman_sandbox_amd as tsa
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
def timed(d, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        tsa.compute_and_apply_rhs(d, st)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
balg = tsa.algorithmic_bytes(4, 72) * 10000
data = tsa.TestData().init_data(10000, 4, 72, device=dev)
timed(data, 150)
for np_, nlev, E in ((4, 72, 12500), (4, 128, 12500), (8, 72, 20000)):
    d = tsa.TestData().init_data(E, np_, nlev, device=dev)
    timed(d, 10)
    del d
    torch.cuda.empty_cache()
print(torch.cuda.memory_summary(abbreviated=True)[:600])
others = [tsa.TestData().init_data(10000, 4, 72, device=dev) for _ in range(5)]
for i, d in enumerate([data] + others):
    timed(d, 40)
    r = [balg / timed(d, 20) / 8e7 for _ in range(3)]
    print("set %d: %s %% of peak; vn0 at 0x%x, v at 0x%x" % (i, " ".join("%.1f" % x for x in r), d.arrays["elem_derived_vn0"].data_ptr(), d.arrays["elem_state_v"].data_ptr()), flush=True)
free, total = torch.cuda.mem_get_info()
print("free %.1f GB of %.1f GB" % (free / 1e9, total / 1e9))
