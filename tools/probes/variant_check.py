"""Full-size A/B of two tuning variants: identical results (bitwise) after several steps, and rates over a cache-window sweep."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, tinman_sandbox_amd as tsa
np_, nlev, E = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
va, vb = int(sys.argv[4]), int(sys.argv[5])
lib = tsa.library().lib
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
balg = tsa.algorithmic_bytes(np_, nlev) * E
da = tsa.TestData().init_data(E, np_, nlev, device=dev)
db = tsa.TestData().init_data(E, np_, nlev, device=dev)
for step in range(3):
    lib.caar_select_variant(np_, nlev, va); tsa.compute_and_apply_rhs(da, st); da.update_time_levels()
    lib.caar_select_variant(np_, nlev, vb); tsa.compute_and_apply_rhs(db, st); db.update_time_levels()
torch.cuda.synchronize()
print("variant %d vs %d after 3 rotating steps on %d elements: " % (va, vb, E) +
      ", ".join("%s %s" % (n[5:], "==" if torch.equal(da.arrays[n], db.arrays[n]) else "max|d|=%.2e" % float((da.arrays[n] - db.arrays[n]).abs().max()))
                for n in tsa.caar.MUTATED))
def timed(d, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(n):
        tsa.compute_and_apply_rhs(d, st)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for w in (0, 64, 128, 160, 192, 208, 224, 240, 256, 288):
    lib.caar_set_cache_window(w << 20)
    out = []
    for v in (va, vb):
        lib.caar_select_variant(np_, nlev, v)
        timed(da, 60)
        out.append(balg / min(timed(da, 30), timed(da, 30)) / 8e7)
    print("window %3d MiB: variant %d %.1f %%   variant %d %.1f %%" % (w, va, out[0], vb, out[1]), flush=True)
lib.caar_set_cache_window(192 << 20)
lib.caar_select_variant(np_, nlev, 0)
