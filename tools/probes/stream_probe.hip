// stream_probe.hip — what the memory system delivers for NR read streams + NW write streams, small short-lived
// workgroups (one 4 KiB row of every stream per 256-thread workgroup, 16 B/lane, nt), as a function of the number
// of streams and the read:write mix.  CAAR reads 13 blocks and writes 8 per element (61:39).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/stream_probe.hip -o tools/probes/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v2 __attribute__((ext_vector_type(2)));
struct Ptrs { const v2* in[16]; v2* out[16]; };

template <int NR, int NW, int THREADS, int ROWS>
__global__ __launch_bounds__(THREADS) void multi(Ptrs p, size_t rows_total) {
  for (int rr = 0; rr < ROWS; ++rr) {
    const size_t i = ((size_t)blockIdx.x * ROWS + rr) * THREADS + threadIdx.x;
    v2 acc = {0, 0};
#pragma unroll
    for (int s = 0; s < NR; ++s) acc += __builtin_nontemporal_load(p.in[s] + i);
#pragma unroll
    for (int s = 0; s < NW; ++s) __builtin_nontemporal_store(acc, p.out[s] + i);
  }
}

template <int NR, int NW, int THREADS, int ROWS>
static void run(const Ptrs& p, size_t n16, const char* what) {
  const size_t blocks = n16 / THREADS / ROWS;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((multi<NR, NW, THREADS, ROWS>), dim3((unsigned)blocks), dim3(THREADS), 0, 0, p, n16);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = (double)(NR + NW) * n16 * 16 * 5;
  printf("%2d read + %2d write streams, %4d-thread WG x %d rows  %7.1f GB/s   %s\n", NR, NW, THREADS, ROWS, bytes / (best * 1e-3) / 1e9, what);
}

int main(int argc, char** argv) {
  const size_t n16 = (size_t)1 << 23;  // 128 MiB per stream
  // argv[1]: byte skew between consecutive streams' bases (default 0: all bases 2 MiB-aligned alike)
  const size_t skew = argc > 1 ? (size_t)atoll(argv[1]) : 0;
  printf("stream base skew: %zu bytes\n", skew);
  Ptrs p;
  for (int s = 0; s < 16; ++s) {
    char *a, *b;
    (void)hipMalloc((void**)&a, n16 * 16 + 64 * skew);
    (void)hipMalloc((void**)&b, n16 * 16 + 64 * skew);
    (void)hipMemset((void*)a, 0, n16 * 16 + 64 * skew);
    p.in[s] = (const v2*)(a + (2 * s) * skew);
    p.out[s] = (v2*)(b + (2 * s + 1) * skew);
  }
  run<4, 1, 256, 1>(p, n16, "many reads, one write");
  run<13, 1, 256, 1>(p, n16, "many reads, one write");
  run<1, 4, 256, 1>(p, n16, "one read, many writes");
  run<1, 8, 256, 1>(p, n16, "one read, many writes");
  run<8, 2, 256, 1>(p, n16, "");
  run<2, 8, 256, 1>(p, n16, "");
  run<1, 1, 256, 1>(p, n16, "copy");
  run<2, 2, 256, 1>(p, n16, "");
  run<4, 4, 256, 1>(p, n16, "");
  run<8, 8, 256, 1>(p, n16, "");
  run<13, 8, 256, 1>(p, n16, "CAAR's mix");
  run<3, 2, 256, 1>(p, n16, "60:40");
  run<2, 1, 256, 1>(p, n16, "67:33");
  run<1, 2, 256, 1>(p, n16, "33:67");
  run<13, 8, 64, 1>(p, n16, "CAAR's mix, one-wave WG");
  run<13, 8, 576, 1>(p, n16, "CAAR's mix, 9-wave WG");
  run<13, 8, 1024, 1>(p, n16, "CAAR's mix, 16-wave WG");
  run<13, 8, 256, 2>(p, n16, "CAAR's mix, 2 rows per WG");
  run<13, 8, 256, 4>(p, n16, "CAAR's mix, 4 rows per WG");
  run<1, 1, 64, 1>(p, n16, "copy, one-wave WG");
  run<1, 1, 1024, 1>(p, n16, "copy, 16-wave WG");
  run<1, 1, 256, 4>(p, n16, "copy, 4 rows per WG (sequential)");
  return 0;
}
