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
//   teardown = control     per-chunk unmap, the range stays RESERVED and the second reservation therefore lands at ANOTHER
//                          address: the arm that must come out clean if the probe itself is sound
//
//   1. reserve a range R of N chunks, create N physical chunks H[j], map H[j] at R + j*chunk, kernel fills R with
//      pattern A (value = A + global index)
//   2. tear down (mode), release H[*], hipMemAddressFree(R) (not in keep_range / control)
//   3. create N new chunks G[j]; map them in plain order into a range F of a size never used before (a fresh address),
//      fill F with pattern B, unmap F: the new chunks now hold B, written through translations nobody doubts
//   4. reserve N chunks again, asking for R's address -> R2 (same address or not: printed); map G[(j+3) % N] at R2 + j*chunk
//      (a rotation: neither R's order nor its reverse — a driver that hands H's physical pages to G in reverse order must
//      not make stale translations look right) and READ R2: with correct translations position j shows that chunk's part of
//      pattern B; a range whose old mappings survived step 2 shows something else (pattern A if H's memory is still as
//      it was).  Nothing is ever WRITTEN through the range under test (round 3's form of this probe did, into memory the
//      driver may have handed to somebody else).
//
//   hipcc --offload-arch=gfx950 -O2 tools/probes/vmm_va_reuse_probe.hip -o tools/probes/vmm_va_reuse_probe
//   for m in control per_chunk single keep_range; do tools/probes/vmm_va_reuse_probe $m; done
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
// position `pos` of the range under test (at p) must show base + (first_index + i); counts what does not, and how much of
// that is the FIRST mapping's pattern (old_base + index in the range)
__global__ void check(const double* p, size_t per_chunk, size_t first_index, double base, size_t pos_index, double old_base,
                      unsigned long long* bad, unsigned long long* old) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < per_chunk; i += (size_t)gridDim.x * blockDim.x) {
    const double v = p[i];
    if (v != base + (double)(first_index + i)) {
      atomicAdd(bad, 1ULL);
      if (v == old_base + (double)(pos_index + i)) atomicAdd(old, 1ULL);
    }
  }
}

int main(int argc, char** argv) {
  const bool control = argc > 1 && std::strcmp(argv[1], "control") == 0;
  const bool keep_range = control || (argc > 1 && std::strcmp(argv[1], "keep_range") == 0);
  const bool per_chunk = keep_range || (argc > 1 && std::strcmp(argv[1], "per_chunk") == 0);
  const char* mode_name = control ? "control" : (keep_range ? "keep_range" : (per_chunk ? "per_chunk" : "single"));
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
              control ? "control (per-chunk unmap, range kept reserved, second mapping at ANOTHER address)"
              : keep_range ? "keep_range (one hipMemUnmap per hipMemMap, range kept reserved and mapped again)"
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
  if (!keep_range) {
    hipError_t e = hipMemAddressFree(R, total);
    std::printf("hipMemAddressFree -> %s\n", hipGetErrorString(e));
    (void)hipGetLastError();
  }

  // 3: the new chunks get pattern B through a fresh range
  void* F = nullptr;
  const size_t fresh_total = total + chunk;  // a size never reserved before: a fresh address
  CK(hipMemAddressReserve(&F, fresh_total, chunk, nullptr, 0));
  for (size_t j = 0; j < N; ++j) CK(hipMemCreate(&G[j], chunk, &prop, 0));
  for (size_t j = 0; j < N; ++j) CK(hipMemMap((char*)F + j * chunk, chunk, 0, G[j], 0));
  CK(hipMemSetAccess(F, total, &acc, 1));
  fill<<<1024, 256>>>((double*)F, total / 8, 2.0e9);   // G[g] holds 2e9 + g*dpc + i
  CK(hipDeviceSynchronize());
  for (size_t j = 0; j < N; ++j) CK(hipMemUnmap((char*)F + j * chunk, chunk));

  // 4: map them into the range under test, in reverse order, and READ
  void* R2 = R;
  if (control || !keep_range) {
    R2 = nullptr;
    CK(hipMemAddressReserve(&R2, total, chunk, control ? nullptr : R, 0));  // (control: R is still reserved, so this is elsewhere)
  }
  std::printf("first range %p, second mapping at %p (%s)\n", R, R2, R == R2 ? "SAME address: the re-use case" : "different address");
  // where did the new chunks' bytes come from?  (G filled through F holds 2e9 + g*dpc + i: a chunk whose memory is H[h]'s old memory
  // cannot be told from here — physical addresses are not visible — so the permutation above must not match any reuse order)
  size_t map_failed = 0;
  for (size_t j = 0; j < N; ++j) {
    hipError_t e = hipMemMap((char*)R2 + j * chunk, chunk, 0, G[(j + 3) % N], 0);
    if (e != hipSuccess) {
      std::printf("hipMemMap of chunk %zu into the range under test -> %s\n", j, hipGetErrorString(e));
      (void)hipGetLastError();
      ++map_failed;
    }
  }
  if (map_failed) {
    std::printf("RESULT %s: %zu of %zu hipMemMap calls into the range under test failed (old mappings still there)\n", mode_name, map_failed, N);
    return 0;
  }
  CK(hipMemSetAccess(R2, total, &acc, 1));
  unsigned long long* cnt_dev;
  CK(hipMalloc((void**)&cnt_dev, 16 * N));
  CK(hipMemset(cnt_dev, 0, 16 * N));
  for (size_t j = 0; j < N; ++j)   // position j shows chunk G[(j+3) % N]: 2e9 + ((j+3) % N)*dpc + i; the first mapping had 1e9 + j*dpc + i there
    check<<<256, 256>>>((const double*)((char*)R2 + j * chunk), dpc, ((j + 3) % N) * dpc, 2.0e9, j * dpc, 1.0e9, cnt_dev + 2 * j, cnt_dev + 2 * j + 1);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> cnt(2 * N);
  CK(hipMemcpy(cnt.data(), cnt_dev, 16 * N, hipMemcpyDeviceToHost));
  unsigned long long all = 0, old = 0;
  for (size_t j = 0; j < N; ++j) {
    if (cnt[2 * j]) std::printf("  position %zu (chunk G[%zu]): %llu of %zu doubles wrong, %llu of them the FIRST mapping's data\n", j, (j + 3) % N, cnt[2 * j], dpc, cnt[2 * j + 1]);
    all += cnt[2 * j];
    old += cnt[2 * j + 1];
  }
  std::printf("RESULT %s: %llu of %zu doubles read through the second mapping are not what the chunks mapped there hold (%llu show the first mapping's data)%s\n",
              mode_name, all, total / 8, old, all ? "  <-- STALE TRANSLATIONS" : " (clean)");
  for (size_t j = 0; j < N; ++j) CK(hipMemUnmap((char*)R2 + j * chunk, chunk));
  for (size_t j = 0; j < N; ++j) CK(hipMemRelease(G[j]));
  CK(hipMemAddressFree(F, fresh_total));
  if (R2 != R) CK(hipMemAddressFree(R2, total));
  if (keep_range) CK(hipMemAddressFree(R, total));
  return 0;
}
