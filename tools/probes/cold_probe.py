import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, tinman_sandbox_amd as tsa
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
t0 = time.perf_counter()
data = tsa.TestData().init_data(10000, 4, 72, device=dev)
torch.cuda.synchronize()
print("init %.3fs" % (time.perf_counter() - t0))
balg = tsa.algorithmic_bytes(4, 72) * 10000
def timed(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        tsa.compute_and_apply_rhs(data, st)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for i in range(12):
    ms = timed(5 if i == 0 else 20)
    print("block %2d (%2d launches): %.4f ms  %.1f%% of 8 TB/s   t=%.3fs" % (i, 5 if i == 0 else 20, ms, balg / ms / 8e7, time.perf_counter() - t0))
