"""Does the steady-state rate depend on WHERE the arrays were allocated?  Same process, same box: allocate the 16
arrays, spin up, measure, free, allocate again (optionally with a dummy allocation in between to shift addresses)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, tinman_sandbox_amd as tsa
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
def steady(data, balg):
    def timed(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(n):
            tsa.compute_and_apply_rhs(data, st)
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    timed(100)
    return [balg / timed(20) / 8e7 for _ in range(4)]
for np_, nlev, E in ((4, 72, 10000), (4, 128, 12500)):
    balg = tsa.algorithmic_bytes(np_, nlev) * E
    keep = []
    for trial in range(8):
        data = tsa.TestData().init_data(E, np_, nlev, device=dev)
        r = steady(data, balg)
        p = data.arrays["elem_state_v"].data_ptr()
        print("np=%d nlev=%d E=%d trial %d: %s %% of peak   state_v at 0x%x" % (np_, nlev, E, trial, " ".join("%.1f" % x for x in r), p), flush=True)
        del data
        if trial % 2 == 1:
            keep.append(torch.empty((trial * 37 + 11) << 20, dtype=torch.uint8, device=dev))  # shift later allocations
        torch.cuda.empty_cache()
    del keep
    torch.cuda.empty_cache()
