// caar_alloc.hip — where the 16 element arrays live in HBM (caar_arrays_alloc / caar_arrays_free; used by caar_create).
//
// Measured on MI355X (DESIGN.md section 5 "Placement", profiles/r02/domain_map.log, spacing_probe.log, vmm_spread_probe.log):
// device memory falls into a few large address classes (stretches of 16-96 GiB), and compute_and_apply_rhs — 21 concurrent
// streams per element — runs 3-5 % faster when its traffic is split over at least two classes than when all arrays lie in
// one (76.2-76.6 % against 73.4-73.9 % of the 8 TB/s peak at NP=4 NLEV=72).  One hipMalloc for everything is always the bad
// case; sixteen hipMallocs are good or bad depending on what the driver hands out.  Physical addresses are not visible, so
// the split is made by construction: every array is backed, through HIP virtual memory management, by physical chunks
// (64 MiB) sampled evenly from a large temporary pool (up to 128 GiB, created chunk by chunk and released again except
// for the chunks kept), mapped into one contiguous virtual range in a scattered order.  Whatever classes the pool covers,
// every stream is spread over them.  Set-up cost ~0.1 s; 18 of 18 fresh processes at the high level.
//
// Small data sets (< 256 MiB), CAAR_PLACEMENT=malloc, or any failure of the VMM route (no support, not enough free
// memory) fall back to one hipMalloc per array.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/caar.h"

struct CaarArena {
  int device;
  bool vmm;
  void* va;        // VMM: the reserved range
  size_t va_bytes;
  std::vector<hipMemGenericAllocationHandle_t>* chunks;  // VMM: the physical chunks kept
  void* plain[CAAR_NUM_ARRAYS];                           // fallback: one hipMalloc per array
  long long pool_chunks, chunk_bytes;
};

namespace {

constexpr size_t kChunk = size_t(64) << 20;
constexpr size_t kSmall = size_t(256) << 20;

size_t pool_limit_bytes() {
  const char* e = std::getenv("CAAR_PLACEMENT_POOL_GIB");
  const long g = e ? std::atol(e) : 128;
  return (g > 0 ? size_t(g) : size_t(128)) << 30;
}
bool want_spread() {
  const char* e = std::getenv("CAAR_PLACEMENT");
  return !(e && std::strcmp(e, "malloc") == 0);
}

void release_vmm(CaarArena* a) {
  if (a->va) {
    (void)hipDeviceSynchronize();  // nothing may still be running on this memory
    (void)hipMemUnmap(a->va, a->va_bytes);
    // The virtual range is deliberately NOT given back (hipMemAddressFree): on ROCm 7.2 / gfx950 a range that is
    // reserved and mapped again later in the same process is still translated to the OLD physical chunks by the GPU
    // (measured: wrong results from the first re-use on, tools/probes/dbg_full.py; none when ranges are never re-used).
    // Address space is not a scarce resource (47 bits); the physical memory is released below.
    a->va = nullptr;
  }
  if (a->chunks) {
    for (hipMemGenericAllocationHandle_t h : *a->chunks) (void)hipMemRelease(h);
    delete a->chunks;
    a->chunks = nullptr;
  }
}

// The VMM route.  Returns false (with everything released) if any step fails.
bool alloc_spread(CaarArena* a, const size_t bytes[CAAR_NUM_ARRAYS], double* out[CAAR_NUM_ARRAYS]) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = a->device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0 ||
      kChunk % gran != 0) {
    (void)hipGetLastError();
    return false;
  }
  size_t total = 0;
  for (int i = 0; i < CAAR_NUM_ARRAYS; ++i) total += (bytes[i] + kChunk - 1) / kChunk * kChunk;
  const size_t need = total / kChunk;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
  size_t pool = pool_limit_bytes();
  if (pool > free_b / 10 * 6) pool = free_b / 10 * 6;  // leave the rest of the device alone
  if (pool < total) return false;
  const size_t pool_chunks = pool / kChunk, stride = pool_chunks / need;
  a->chunks = new (std::nothrow) std::vector<hipMemGenericAllocationHandle_t>();
  if (!a->chunks) return false;
  a->chunks->reserve(need);
  // create the pool chunk by chunk; keep every stride-th chunk, give the others back at the end (releasing a chunk
  // at once would let the next hipMemCreate return the same memory and the pool would cover nothing)
  std::vector<hipMemGenericAllocationHandle_t> spare;
  spare.reserve(pool_chunks);
  bool ok = true;
  for (size_t c = 0; c < pool_chunks && ok; ++c) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, kChunk, &prop, 0) != hipSuccess) {
      (void)hipGetLastError();
      ok = a->chunks->size() == need;  // the device filled up: fine if we already have what we need
      break;
    }
    if (c % stride == 0 && a->chunks->size() < need) a->chunks->push_back(h);
    else spare.push_back(h);
  }
  while (ok && a->chunks->size() < need && !spare.empty()) {  // (pool cut short) top up from the spare chunks
    a->chunks->push_back(spare.back());
    spare.pop_back();
  }
  for (hipMemGenericAllocationHandle_t h : spare) (void)hipMemRelease(h);
  ok = ok && a->chunks->size() == need;
  if (ok) {
    if (hipMemAddressReserve(&a->va, total, kChunk, nullptr, 0) != hipSuccess) {  // chunk-aligned if the driver agrees
      (void)hipGetLastError();
      a->va = nullptr;
      ok = hipMemAddressReserve(&a->va, total, 0, nullptr, 0) == hipSuccess;
    }
  }
  if (ok) {
    a->va_bytes = total;
    // virtual chunk j <- kept chunk (j * 7) mod need (7 is coprime to any need that is not a multiple of 7; if it is, use 11):
    // neighbouring 64 MiB pieces of an array come from distant parts of the pool
    const size_t mul = need % 7 ? 7 : (need % 11 ? 11 : 1);
    for (size_t j = 0; j < need && ok; ++j)
      ok = hipMemMap((char*)a->va + j * kChunk, kChunk, 0, (*a->chunks)[(j * mul) % need], 0) == hipSuccess;
  }
  if (ok) {
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    ok = hipMemSetAccess(a->va, total, &acc, 1) == hipSuccess;
  }
  if (!ok) {
    (void)hipGetLastError();
    release_vmm(a);
    return false;
  }
  size_t off = 0;
  for (int i = 0; i < CAAR_NUM_ARRAYS; ++i) {
    out[i] = reinterpret_cast<double*>((char*)a->va + off);
    off += (bytes[i] + kChunk - 1) / kChunk * kChunk;
  }
  a->vmm = true;
  a->pool_chunks = (long long)pool_chunks;
  a->chunk_bytes = (long long)kChunk;
  return true;
}

}  // namespace

extern "C" {

int caar_arrays_alloc(CaarArena** arena, const CaarDims* dims, int device, CaarArrays* out_dev) {
  if (!arena || !dims || !out_dev || dims->num_elems <= 0 || dims->qsize_d < 1 || dims->timelevels < 1 || dims->np < 1 ||
      dims->nlev < 1)
    return CAAR_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CAAR_ENODEVICE;
  if (device < 0 || device >= ndev) return CAAR_EINVAL;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return (int)e;
  CaarArena* a = new (std::nothrow) CaarArena();
  if (!a) return CAAR_ENOMEM;
  a->device = device;
  a->vmm = false;
  a->va = nullptr;
  a->va_bytes = 0;
  a->chunks = nullptr;
  a->pool_chunks = a->chunk_bytes = 0;
  std::memset(a->plain, 0, sizeof(a->plain));
  size_t bytes[CAAR_NUM_ARRAYS], total = 0;
  for (int i = 0; i < CAAR_NUM_ARRAYS; ++i) {
    bytes[i] = sizeof(double) * (size_t)caar_array_len(dims, i);
    total += bytes[i];
  }
  double** out = reinterpret_cast<double**>(out_dev);
  if (!(want_spread() && total >= kSmall && alloc_spread(a, bytes, out))) {
    for (int i = 0; i < CAAR_NUM_ARRAYS; ++i) {
      e = hipMalloc(&a->plain[i], bytes[i]);
      if (e != hipSuccess) {
        (void)caar_arrays_free(a);
        return e == hipErrorOutOfMemory ? CAAR_ENOMEM : (int)e;
      }
      out[i] = static_cast<double*>(a->plain[i]);
    }
  }
  *arena = a;
  return CAAR_OK;
}

int caar_arrays_free(CaarArena* a) {
  if (!a) return CAAR_EINVAL;
  (void)hipSetDevice(a->device);
  release_vmm(a);
  for (int i = 0; i < CAAR_NUM_ARRAYS; ++i)
    if (a->plain[i]) (void)hipFree(a->plain[i]);
  delete a;
  return CAAR_OK;
}

int caar_arrays_placement(const CaarArena* a, long long* pool_chunks, long long* chunk_bytes) {
  if (!a) return CAAR_EINVAL;
  if (pool_chunks) *pool_chunks = a->pool_chunks;
  if (chunk_bytes) *chunk_bytes = a->chunk_bytes;
  return a->vmm ? 1 : 0;
}

}  // extern "C"
