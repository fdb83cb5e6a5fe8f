"""Bandwidth map of one huge allocation: the tuned copy inside consecutive 512 MiB windows (first half -> second half).
If the allocation-to-allocation spread of the kernels is a property of physical regions, the map shows it."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, tinman_sandbox_amd as tsa
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev)
L = tsa.library()
sv = C.c_void_p(st.cuda_stream)
GB = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = 512 << 20
big = torch.zeros(GB << 27, dtype=torch.float64, device=dev)   # GB GiB
base = big.data_ptr()
n = W // 2 // 8
def bw(off):
    src, dst = C.c_void_p(base + off), C.c_void_p(base + off + W // 2)
    f = lambda: L.check(L.lib.caar_stream_copy_tuned(dst, src, n, 15, sv), "c")
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(20):
        f()
    e1.record(st)
    torch.cuda.synchronize()
    return 2 * n * 8 * 20 / e0.elapsed_time(e1) / 1e6
for _ in range(300):
    L.lib.caar_stream_copy_tuned(C.c_void_p(base + W // 2), C.c_void_p(base), n, 15, sv)
torch.cuda.synchronize()
print("base 0x%x, %d GiB, %d windows of 512 MiB: copy GB/s per window" % (base, GB, (GB << 30) // W))
vals = [bw(i * W) for i in range((GB << 30) // W)]
for i in range(0, len(vals), 16):
    print("  %3d GiB: " % (i // 2) + " ".join("%4.0f" % v for v in vals[i:i + 16]))
vals2 = [bw(i * W) for i in range(0, min(32, len(vals)))]
print("  again, first 32 windows: " + " ".join("%4.0f" % v for v in vals2))
