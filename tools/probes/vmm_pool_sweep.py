"""How large must the temporary pool of caar_arrays_alloc_ex be?  (DESIGN.md section 5 "Placement")

For each pool bound (GiB) a number of FRESH processes allocate the 10 000-element NP=4 NLEV=72 data set through the library
(CaarPlacement.pool_bytes), spin the default kernel up and report the steady rate with the cache window and all-streaming.
The default bound (CAAR_PLACEMENT_POOL_DEFAULT) is the smallest one whose processes all reach the high level.

    python tools/probes/vmm_pool_sweep.py [--procs 8] [--pools 0,2,4,8,16,32,64,128]      (0 = plain hipMalloc per array)
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(pool_gib, elems, np_, nlev):
    import torch
    import tinman_sandbox_amd as tsa
    lib = tsa.library().lib
    dev = torch.device("cuda", 0)
    place = tsa.placement("malloc") if pool_gib == 0 else tsa.placement("spread", pool_gib=pool_gib, max_free_fraction=0.9)
    t0 = time.time()
    data = tsa.TestData().init_data(elems, np_, nlev, device=dev, place=place)
    torch.cuda.synchronize()
    t_alloc = time.time() - t0
    st = torch.cuda.current_stream(dev)

    def timed(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(n):
            tsa.compute_and_apply_rhs(data, st)
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    balg = tsa.algorithmic_bytes(np_, nlev) * elems
    timed(150)
    hyb = min(timed(20) for _ in range(3))
    lib.caar_select_variant(np_, nlev, 1)
    timed(40)
    nt = min(timed(20) for _ in range(3))
    print("POOL %g GiB used %.1f GiB alloc+init %.2f s  window %.2f %%  all-streaming %.2f %%" % (
        pool_gib, data.arrays.arena.pool_bytes() / 2 ** 30, t_alloc, balg / hyb / 8e7, balg / nt / 8e7), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", type=float, default=None)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--pools", default="0,2,4,8,16,32,64,128")
    ap.add_argument("--elems", type=int, default=10000)
    ap.add_argument("--np", type=int, default=4, dest="np_")
    ap.add_argument("--nlev", type=int, default=72)
    a = ap.parse_args()
    if a.child is not None:
        return child(a.child, a.elems, a.np_, a.nlev)
    pools = [float(x) for x in a.pools.split(",")]
    for rnd in range(a.procs):  # round-robin over the bounds so that drift of the box hits all of them alike
        for g in pools:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(g), "--elems", str(a.elems),
                                "--np", str(a.np_), "--nlev", str(a.nlev)], capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith("POOL")]
            print(line[0] if line else "POOL %g GiB FAILED: %s" % (g, (r.stderr or r.stdout)[-300:]), flush=True)


if __name__ == "__main__":
    main()
