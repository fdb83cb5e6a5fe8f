#include <hip/hip_runtime_api.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
int main() {
  CK(hipSetDevice(0));
  void* warm; CK(hipMalloc(&warm, 1 << 20));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
  for (int rep = 0; rep < 3; ++rep)
    for (size_t chunk_mib : {64, 1024, 4096, 16384}) {
      const size_t chunk = chunk_mib << 20, n = (size_t(128) << 30) / chunk;
      std::vector<hipMemGenericAllocationHandle_t> h(n);
      auto t0 = std::chrono::steady_clock::now();
      for (size_t i = 0; i < n; ++i) CK(hipMemCreate(&h[i], chunk, &prop, 0));
      auto t1 = std::chrono::steady_clock::now();
      for (size_t i = 0; i < n; ++i) CK(hipMemRelease(h[i]));
      auto t2 = std::chrono::steady_clock::now();
      std::printf("rep %d: 128 GiB as %5zu x %5zu MiB: create %.3f s, release %.3f s\n", rep, n, chunk_mib,
                  std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count());
    }
  return 0;
}
