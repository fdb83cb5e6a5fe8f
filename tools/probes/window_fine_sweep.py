import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tinman_sandbox_amd as tsa
lib = tsa.library().lib
lib.caar_set_adaptive_window(0)
def timed(data, n=40):
    for _ in range(8): tsa.compute_and_apply_rhs(data)
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): tsa.compute_and_apply_rhs(data)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best
for nlev, E in ((72, 10000), (72, 12500), (72, 15000), (72, 20000), (128, 12500), (128, 8000)):
    data = tsa.TestData().init_data(E, 4, nlev, device="cuda")
    for rep in range(1):
        row = []
        for mb in (176, 192, 200, 208, 216, 224, 232, 240):
            lib.caar_set_cache_window(mb << 20)
            row.append("%dMB %.4f" % (mb, timed(data)))
        print("nlev=%d E=%d: " % (nlev, E) + " | ".join(row), flush=True)
    del data; torch.cuda.empty_cache()
lib.caar_set_cache_window(224 << 20)
