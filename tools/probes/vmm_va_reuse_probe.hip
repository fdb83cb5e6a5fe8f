// vmm_va_reuse_probe.hip — bare HIP, no library: what happens to a virtual range that was mapped chunk by chunk,
// torn down, and reserved + mapped again in the same process?
//
// Round 2 saw wrong results after an arena of csrc/caar_alloc.hip was freed and a new one mapped, blamed "stale GPU
// translations on ROCm 7.2" and stopped giving ranges back.  The arena had been unmapped with ONE hipMemUnmap over the
// whole range although it had been mapped by one hipMemMap per 64 MiB chunk.  This probe separates the two candidates:
//
//   teardown = single      hipMemUnmap(va, total) once, hipMemAddressFree            (what round 2's release_vmm did first)
//   teardown = per_chunk   hipMemUnmap(va + j*chunk, chunk) each, hipMemAddressFree  (mirror of the hipMemMap loop)
//   teardown = keep_range  per-chunk unmap, the range is NOT freed and is mapped again as it is
//
//   1. reserve a range R of N chunks, create N physical chunks H[j], map H[j] at R + j*chunk, kernel fills R with
//      pattern A (value = A + global index)
//   2. tear down (mode), release H[*], hipMemAddressFree(R)
//   3. reserve N chunks again, asking for R's address -> R2 (same address or not: printed), create N new chunks G[j], map G[N-1-j] at
//      R2 + j*chunk (another order than before), kernel fills R2 with pattern B
//   4. unmap R2 chunk by chunk; map G[j] in plain order into a range F of a size never used before (a fresh address),
//      and count the doubles of F that hold what step 3 must have left there.  With correct translations every double
//      matches; a range whose old mappings survived step 2 sends step 3's writes to H's (released) memory instead.
//
//   hipcc --offload-arch=gfx950 -O2 tools/probes/vmm_va_reuse_probe.hip -o tools/probes/vmm_va_reuse_probe
//   tools/probes/vmm_va_reuse_probe single ; tools/probes/vmm_va_reuse_probe per_chunk
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                            \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      std::printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);     \
      std::exit(2);                                                                      \
    }                                                                                    \
  } while (0)

__global__ void fill(double* p, size_t n, double base) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = base + (double)i;
}
// counts doubles of physical chunk g (mapped at p) that do not hold base + (index the chunk had in the reused range)
__global__ void check(const double* p, size_t per_chunk, size_t first_index, double base, unsigned long long* bad) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per_chunk; i += (size_t)gridDim.x * blockDim.x)
    if (p[i] != base + (double)(first_index + i)) atomicAdd(bad, 1ULL);
}

int main(int argc, char** argv) {
  const bool keep_range = argc > 1 && std::strcmp(argv[1], "keep_range") == 0;
  const bool per_chunk = keep_range || (argc > 1 && std::strcmp(argv[1], "per_chunk") == 0);
  const char* mode_name = keep_range ? "keep_range" : (per_chunk ? "per_chunk" : "single");
  const size_t N = argc > 2 ? (size_t)std::atoi(argv[2]) : 8;
  const size_t chunk = size_t(64) << 20, total = N * chunk, dpc = chunk / 8;
  CK(hipSetDevice(0));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  std::printf("teardown mode: %s, %zu chunks of 64 MiB\n",
              keep_range ? "keep_range (one hipMemUnmap per hipMemMap, range kept reserved)"
                         : (per_chunk ? "per_chunk (one hipMemUnmap per hipMemMap, hipMemAddressFree)" : "single (one hipMemUnmap over the range, hipMemAddressFree)"), N);

  // 1
  void* R = nullptr;
  CK(hipMemAddressReserve(&R, total, chunk, nullptr, 0));
  std::vector<hipMemGenericAllocationHandle_t> H(N), G(N);
  for (size_t j = 0; j < N; ++j) CK(hipMemCreate(&H[j], chunk, &prop, 0));
  for (size_t j = 0; j < N; ++j) CK(hipMemMap((char*)R + j * chunk, chunk, 0, H[j], 0));
  CK(hipMemSetAccess(R, total, &acc, 1));
  fill<<<1024, 256>>>((double*)R, total / 8, 1.0e9);
  CK(hipDeviceSynchronize());

  // 2
  if (per_chunk) {
    for (size_t j = 0; j < N; ++j) CK(hipMemUnmap((char*)R + j * chunk, chunk));
  } else {
    hipError_t e = hipMemUnmap(R, total);
    std::printf("hipMemUnmap(range, total) -> %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
  }
  for (size_t j = 0; j < N; ++j) {
    hipError_t e = hipMemRelease(H[j]);
    if (e != hipSuccess) std::printf("hipMemRelease(H[%zu]) -> %s\n", j, hipGetErrorString(e));
  }
  void* R2 = R;
  if (!keep_range) {
    hipError_t e = hipMemAddressFree(R, total);
    std::printf("hipMemAddressFree -> %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
    // 3
    R2 = nullptr;
    CK(hipMemAddressReserve(&R2, total, chunk, R, 0));  // ask for the address just given back
  }
  std::printf("first range %p, second reservation %p (%s)\n", R, R2, R == R2 ? "SAME address: the re-use case" : "different address");
  for (size_t j = 0; j < N; ++j) CK(hipMemCreate(&G[j], chunk, &prop, 0));
  size_t map_failed = 0;
  for (size_t j = 0; j < N; ++j) {
    hipError_t e = hipMemMap((char*)R2 + j * chunk, chunk, 0, G[N - 1 - j], 0);
    if (e != hipSuccess) {
      std::printf("hipMemMap of chunk %zu into the re-used range -> %s\n", j, hipGetErrorString(e));
      (void)hipGetLastError();
      ++map_failed;
    }
  }
  if (map_failed) {
    std::printf("RESULT %s: %zu of %zu hipMemMap calls into the re-used range failed (old mappings still there)\n",
                mode_name, map_failed, N);
    return 0;
  }
  CK(hipMemSetAccess(R2, total, &acc, 1));
  fill<<<1024, 256>>>((double*)R2, total / 8, 2.0e9);
  CK(hipDeviceSynchronize());

  // 4
  for (size_t j = 0; j < N; ++j) CK(hipMemUnmap((char*)R2 + j * chunk, chunk));
  void* F = nullptr;
  const size_t fresh_total = total + chunk;  // a size never reserved before: a fresh address
  CK(hipMemAddressReserve(&F, fresh_total, chunk, nullptr, 0));
  for (size_t j = 0; j < N; ++j) CK(hipMemMap((char*)F + j * chunk, chunk, 0, G[j], 0));
  CK(hipMemSetAccess(F, total, &acc, 1));
  unsigned long long* bad_dev;
  CK(hipMalloc((void**)&bad_dev, 8 * N));
  CK(hipMemset(bad_dev, 0, 8 * N));
  // G[g] was mapped at position N-1-g of the re-used range
  for (size_t g = 0; g < N; ++g)
    check<<<256, 256>>>((const double*)((char*)F + g * chunk), dpc, (N - 1 - g) * dpc, 2.0e9, bad_dev + g);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> bad(N);
  CK(hipMemcpy(bad.data(), bad_dev, 8 * N, hipMemcpyDeviceToHost));
  unsigned long long all = 0;
  for (size_t g = 0; g < N; ++g) {
    if (bad[g]) std::printf("  physical chunk %zu (position %zu of the re-used range): %llu of %zu doubles wrong\n", g, N - 1 - g, bad[g], dpc);
    all += bad[g];
  }
  std::printf("RESULT %s: %llu of %zu doubles written through the re-used range did not reach the chunks mapped there%s\n",
              mode_name, all, total / 8, all ? "  <-- STALE TRANSLATIONS" : " (clean)");
  for (size_t j = 0; j < N; ++j) CK(hipMemUnmap((char*)F + j * chunk, chunk));
  for (size_t j = 0; j < N; ++j) CK(hipMemRelease(G[j]));
  CK(hipMemAddressFree(F, fresh_total));
  CK(hipMemAddressFree(R2, total));
  return 0;
}
