// simd_placement_probe.hip — which SIMD of its CU each wave of a workgroup lands on (gfx950).
//
// The NP=4 NLEV=72 kernels run two 3-wave workgroups per CU.  A CU has four SIMDs, so six
// waves cannot be spread evenly: this probe records HW_REG_HW_ID (gfx9 layout: wave_id [3:0], simd_id [5:4], cu_id [11:8],
// sh_id [12], se_id [15:13]) and HW_REG_XCC_ID of every wave of a launch that is shaped like the kernel's (workgroup
// size, LDS per workgroup, many more workgroups than fit at once, each busy for a while) and prints, per workgroup
// size, how the waves of a workgroup are placed and how many waves the busiest SIMD of a CU holds when the CU is full.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/simd_placement_probe.hip -o tools/probes/simd_placement_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

struct Rec {
  unsigned hw_id, xcc_id;
  unsigned long long t0, t1;
};

__global__ void probe(Rec* out, int spin) {
  extern __shared__ double lds[];
  const int wave = threadIdx.x >> 6;
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = __builtin_readcyclecounter();
  double x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = __builtin_fma(x, 1.0000001, 0.5);  // keep the wave resident for a while
  lds[threadIdx.x] = x;
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0) {
    Rec r;
    r.hw_id = hw;
    r.xcc_id = xcc + (lds[threadIdx.x ^ 1] == -1.0 ? 1 : 0);
    r.t0 = t0;
    r.t1 = t1;
    out[(size_t)blockIdx.x * (blockDim.x >> 6) + wave] = r;
  }
}

int main() {
  CHECK(hipSetDevice(0));
  struct Shape {
    int waves, lds_bytes;
    const char* what;
  } shapes[] = {{3, 73248, "3 waves, 73 KB (NLEV=72 step loop: two per CU)"},
                {3, 9216, "3 waves, 9 KB (NLEV=72 single call, MINW = 1)"},
                {4, 73248, "4 waves, 73 KB (two per CU)"},
                {6, 73248, "6 waves, 73 KB (two per CU)"},
                {6, 101000, "6 waves, 101 KB (one per CU)"},
                {8, 130560, "8 waves, 130 KB (NP=8: one per CU)"}};
  const int blocks = 4096, spin = 20000;
  for (const Shape& s : shapes) {
    Rec* d = nullptr;
    const size_t n = (size_t)blocks * s.waves;
    CHECK(hipMalloc(&d, n * sizeof(Rec)));
    CHECK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, s.lds_bytes));
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(s.waves * 64), s.lds_bytes, 0, d, spin);
    CHECK(hipDeviceSynchronize());
    std::vector<Rec> h(n);
    CHECK(hipMemcpy(h.data(), d, n * sizeof(Rec), hipMemcpyDeviceToHost));
    CHECK(hipFree(d));
    // (1) the SIMD sequence of a workgroup's waves
    std::map<std::vector<int>, int> patterns;
    for (int b = 0; b < blocks; ++b) {
      std::vector<int> p;
      for (int w = 0; w < s.waves; ++w) p.push_back((h[(size_t)b * s.waves + w].hw_id >> 4) & 3);
      ++patterns[p];
    }
    std::printf("== %s\n   SIMD of wave 0, 1, ... of a workgroup (count of workgroups):\n", s.what);
    int shown = 0;
    for (auto& kv : patterns) {
      if (shown++ >= 12) {
        std::printf("     ... %zu patterns in all\n", patterns.size());
        break;
      }
      std::printf("     ");
      for (int x : kv.first) std::printf("%d ", x);
      std::printf(": %d\n", kv.second);
    }
    // (2) waves per SIMD of a CU while the CU is full: for every workgroup, the waves (of any workgroup) on the same
    // CU whose residency interval covers this workgroup's midpoint
    std::map<unsigned, std::vector<size_t>> by_cu;
    for (size_t i = 0; i < n; ++i) {
      const unsigned cu = ((h[i].hw_id >> 8) & 0xff) | (h[i].xcc_id << 8);  // cu_id, sh_id, se_id + XCC
      by_cu[cu].push_back(i);
    }
    std::map<std::vector<int>, int> loads;
    for (auto& kv : by_cu) {
      const std::vector<size_t>& idx = kv.second;
      for (size_t a : idx) {
        if ((a % s.waves) != 0) continue;  // one sample per workgroup
        const unsigned long long mid = (h[a].t0 + h[a].t1) / 2;
        std::vector<int> per(4, 0);
        for (size_t b : idx)
          if (h[b].t0 <= mid && mid <= h[b].t1) ++per[(h[b].hw_id >> 4) & 3];
        ++loads[per];
      }
    }
    // (3) for two workgroups on a CU: where the other one's wave 0 sits relative to this one's
    std::map<int, int> rel;
    for (auto& kv : by_cu) {
      const std::vector<size_t>& idx = kv.second;
      for (size_t a : idx) {
        if ((a % s.waves) != 0) continue;
        const unsigned long long mid = (h[a].t0 + h[a].t1) / 2;
        for (size_t b : idx)
          if (b != a && (b % s.waves) == 0 && h[b].t0 <= mid && mid <= h[b].t1)
            ++rel[(int)((((h[b].hw_id >> 4) & 3) - ((h[a].hw_id >> 4) & 3)) & 3)];
      }
    }
    std::printf("   SIMD of the co-resident workgroups' wave 0 minus this one's, mod 4 (count):");
    for (auto& kv : rel) std::printf("  %+d: %d", kv.first, kv.second);
    std::printf("\n");
    std::printf("   waves on SIMD 0..3 of a CU at a workgroup's midpoint (count of samples), %zu CUs seen:\n", by_cu.size());
    std::vector<std::pair<int, std::vector<int>>> order;
    for (auto& kv : loads) order.push_back({kv.second, kv.first});
    std::sort(order.rbegin(), order.rend());
    for (size_t i = 0; i < order.size() && i < 8; ++i)
      std::printf("     %d %d %d %d : %d\n", order[i].second[0], order[i].second[1], order[i].second[2], order[i].second[3], order[i].first);
  }
  return 0;
}
