// vmm_spread_probe.cpp — can the address-class effect (DESIGN.md section 5 "Placement") be removed by construction?
// The 16 arrays are backed through HIP virtual memory management: physical chunks are drawn from a large temporary
// pool (every k-th chunk kept, the rest released), so every array is spread over the physical regions the pool covers.
//   build: see tools/probes/README (g++ ... -lhomme_caar -lcaar_hip -lamdhip64)
//   run:   vmm_spread_probe <mode> [chunk MiB] [pool GiB]     mode: malloc | slab | vmm_seq | vmm_spread
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "caar.h"
#include "homme_data.hpp"

namespace Homme { int num_elems = 0; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "malloc";
  const size_t chunk = (argc > 2 ? std::atoi(argv[2]) : 2) * (size_t(1) << 20);
  const size_t pool_bytes = (argc > 3 ? std::atoi(argv[3]) : 64) * (size_t(1) << 30);
  const int E = 10000;
  Homme::num_elems = E;
  Homme::TestData data;
  data.init_data();
  CK(hipSetDevice(0));
  CaarDims dims = {Homme::np, Homme::nlev, Homme::qsize_d, Homme::timelevels, E};
  size_t bytes[16], total = 0;
  for (int i = 0; i < 16; ++i) {
    bytes[i] = size_t(caar_array_len(&dims, i)) * 8;
    total += (bytes[i] + chunk - 1) / chunk * chunk;
  }
  double* dev[16];
  double** host = reinterpret_cast<double**>(&data.arrays);
  if (mode == "malloc") {
    for (int i = 0; i < 16; ++i) CK(hipMalloc((void**)&dev[i], bytes[i]));
  } else if (mode == "slab") {
    char* slab;
    CK(hipMalloc((void**)&slab, total));
    size_t off = 0;
    for (int i = 0; i < 16; ++i) { dev[i] = (double*)(slab + off); off += (bytes[i] + chunk - 1) / chunk * chunk; }
  } else {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    if (chunk % gran) { std::printf("chunk must be a multiple of %zu\n", gran); return 2; }
    const size_t need = total / chunk;
    const size_t pool_chunks = mode == "vmm_spread" ? pool_bytes / chunk : need;
    const size_t stride = pool_chunks / need;
    auto t0 = std::chrono::steady_clock::now();
    std::vector<hipMemGenericAllocationHandle_t> h(pool_chunks);
    for (size_t c = 0; c < pool_chunks; ++c) CK(hipMemCreate(&h[c], chunk, &prop, 0));
    std::vector<hipMemGenericAllocationHandle_t> keep;
    for (size_t c = 0; c < pool_chunks; ++c) {
      if (c % stride == 0 && keep.size() < need) keep.push_back(h[c]);
      else CK(hipMemRelease(h[c]));
    }
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, total, 0, nullptr, 0));
    // chunk j of the VA range <- kept chunk (j * 7) mod need: neighbouring pages of an array come from distant places
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (size_t j = 0; j < need; ++j) CK(hipMemMap((char*)va + j * chunk, chunk, 0, keep[mode == "vmm_spread" ? (j * 7) % need : j], 0));
    CK(hipMemSetAccess(va, total, &acc, 1));
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("vmm: granularity %zu, %zu chunks of %zu MiB kept of %zu created (pool %.1f GiB), set-up %.3f s\n", gran, need,
                chunk >> 20, pool_chunks, pool_chunks * chunk / 1073741824.0, sec);
    size_t off = 0;
    for (int i = 0; i < 16; ++i) { dev[i] = (double*)((char*)va + off); off += (bytes[i] + chunk - 1) / chunk * chunk; }
  }
  for (int i = 0; i < 16; ++i) CK(hipMemcpy(dev[i], host[i], bytes[i], hipMemcpyHostToDevice));
  CaarArrays a;
  std::memcpy(&a, dev, sizeof(a));
  double* dvv;
  CK(hipMalloc((void**)&dvv, 16 * 8));
  CK(hipMemcpy(dvv, &data.deriv.Dvv[0][0], 16 * 8, hipMemcpyHostToDevice));
  CaarParams p = {};
  p.nets = 0; p.nete = E; p.n0 = 0; p.np1 = 1; p.nm1 = 2; p.qn0 = 0; p.dt2 = 1.0;
  p.rrearth = data.constants.rrearth; p.eta_ave_w = 1.0; p.Rwater_vapor = 461.5; p.Rgas = 287.04; p.kappa = 287.04 / 1005.0;
  p.ps0 = data.hvcoord.ps0; p.hyai0 = data.hvcoord.hyai[0]; p.Dvv = &data.deriv.Dvv[0][0]; p.rsplit = 1;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](int n) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < n; ++i) if (caar_launch(&dims, &a, dvv, &p, st) != 0) { std::printf("launch failed\n"); std::exit(3); }
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / n;
  };
  run(150);
  const double balg = double(caar_algorithmic_bytes(4, 72, 0)) * E;
  std::printf("%-10s chunk %3zu MiB: %.1f %.1f %.1f %% of 8 TB/s\n", mode.c_str(), chunk >> 20, balg / run(20) / 8e7, balg / run(20) / 8e7,
              balg / run(20) / 8e7);
  return 0;
}
