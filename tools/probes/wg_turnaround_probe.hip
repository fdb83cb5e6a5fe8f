// wg_turnaround_probe.hip — what does it cost to replace a finished workgroup by the next one?
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/wg_turnaround_probe.hip -o tools/probes/wg_turnaround_probe && ./wg_turnaround_probe
//
// The CAAR kernels run one workgroup per element, two resident per CU (512 slots), 10 000 elements per launch: every slot
// is refilled ~19.5 times per launch.  If refilling a slot costs a microsecond, that is 6 % of a 0.3 ms launch that a
// persistent kernel would not pay.  Each workgroup here (256 threads, 72 KB of LDS so that exactly two fit a CU) does nothing
// but wait X microseconds on the wall clock; launches of R x 512 workgroups then take R * X plus R times the turnaround.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void wait_kernel(long long ticks, unsigned* sink) {
  __shared__ unsigned lds[72 * 1024 / 4];
  const long long t0 = wall_clock64();
  lds[threadIdx.x] = threadIdx.x;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (lds[(threadIdx.x + 1) & 255] == 0xffffffffu) *sink = 1;  // keeps the LDS allocation alive
}

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                 \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

int main() {
  int rate_khz = 0;
  CHECK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int slots = p.multiProcessorCount * 2;
  unsigned* sink;
  CHECK(hipMalloc(&sink, 4));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  std::printf("%d CUs, %d slots, wall clock %d kHz\n", p.multiProcessorCount, slots, rate_khz);
  for (double us : {2.0, 5.0, 10.0, 15.0}) {
    const long long ticks = (long long)(us * 1e-3 * rate_khz);
    for (int rounds : {1, 20}) {
      const int grid = rounds * slots;
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(wait_kernel, dim3(grid), dim3(256), 0, 0, ticks, sink);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
      }
      std::printf("wait %5.1f us  rounds %2d (%5d workgroups)  %8.2f us  = %6.3f us per round beyond the wait\n", us, rounds, grid,
                  best * 1e3, (best * 1e3 - rounds * us) / rounds);
    }
  }
  return 0;
}
