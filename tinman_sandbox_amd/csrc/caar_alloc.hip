// caar_alloc.hip — where the 16 element arrays live in HBM (caar_arrays_alloc[_ex] / caar_arrays_free; used by caar_create).
//
// Measured on MI355X (DESIGN.md section 5 "Placement", profiles/r02/domain_map.log, spacing_probe.log, vmm_spread_probe.log):
// device memory falls into a few large address classes (stretches of 16-96 GiB), and compute_and_apply_rhs — 21 concurrent
// streams per element — runs 3-5 % faster when its traffic is split over at least two classes than when all arrays lie in
// one (76.2-76.6 % against 73.4-73.9 % of the 8 TB/s peak at NP=4 NLEV=72).  One hipMalloc for everything is always the bad
// case; sixteen hipMallocs are good or bad depending on what the driver hands out.  Physical addresses are not visible, so
// the split is made by construction: every array is backed, through HIP virtual memory management, by physical chunks
// (64 MiB) sampled evenly from a temporary pool that is created chunk by chunk and released again except for the chunks
// kept, mapped into one contiguous virtual range in a scattered order.  Whatever classes the pool covers, every stream is
// spread over them.
//
// The pool is a transient claim on device memory, so it is bounded three ways (CaarPlacement, include/caar.h):
//   * by an absolute size (default kDefaultPoolBytes; profiles/r03/vmm_pool_sweep.log is the sweep behind the number),
//   * by a fraction of the memory that is FREE when the call is made (default one half: a neighbour — another rank on the
//     same GPU, torch's allocator — keeps at least the other half),
//   * and it is never an error: if the device fills up while the pool is being created the spare chunks are released
//     first and the arrays are topped up from what is then free; if that fails too, one hipMalloc per array.
// Small data sets (< 256 MiB), policy CAAR_PLACE_MALLOC (or CAAR_PLACEMENT=malloc in the environment when the caller did
// not choose) fall back to one hipMalloc per array as well.
//
// Teardown mirrors set-up: every chunk was mapped by its own hipMemMap, so every chunk is unmapped by its own
// hipMemUnmap, return codes checked, the physical chunks released.  The VIRTUAL range is not given back
// (hipMemAddressFree): a conservative policy, NOT a demonstrated driver defect.  Round 2 saw wrong results after an arena
// had been freed (with ONE hipMemUnmap over a range mapped chunk by chunk) and a new one mapped; round 3's bare-HIP probe
// wrote through a re-reserved range and read back through a fresh one and called every double stale — in all of its arms,
// so the probe itself could not be told from the driver (no control arm).  Round 4's form of the probe
// (tools/probes/vmm_va_reuse_probe.hip: new chunks filled through a fresh range, mapped into the range under test in a
// rotated order, READ there; control arm = second mapping at another address) finds every arm clean, same-address re-use
// included, whether the range was unmapped by one call or chunk by chunk (profiles/r04/vmm_va_reuse_probe.log).  What round
// 2 saw is therefore unexplained rather than pinned on the driver, and the policy stays because it cannot be wrong:
// address space is not scarce (a 10 000-element arena takes 2 GiB of 128 TiB); the physical memory is what is returned.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/caar.h"
#include "../../include/caar_tuning.h"

struct CaarArena {
  int device;
  bool vmm;
  void* va;        // VMM: the reserved range
  size_t va_bytes;
  std::vector<hipMemGenericAllocationHandle_t>* chunks;  // VMM: the physical chunks kept, in VIRTUAL order (chunk j backs va + j * kChunk)
  size_t mapped;                                          // VMM: how many of them are mapped
  void* plain[CAAR_NUM_ARRAYS];                           // fallback: one hipMalloc per array
  long long pool_chunks, chunk_bytes;
  int teardown_errors;  // hipMemUnmap / hipMemRelease / hipMemAddressFree calls that failed (reported by caar_arrays_free)
  const void* vn0;      // elem_derived_vn0 of the set: the key of its adaptive-window tuner (caar_abi.hip), dropped with the arena
};

namespace caar {
void window_tuner_forget(const void* key, int device);  // caar_abi.hip
}

namespace {

constexpr size_t kChunk = size_t(64) << 20;
constexpr size_t kSmall = size_t(256) << 20;
constexpr long long kDefaultPoolBytes = CAAR_PLACEMENT_POOL_DEFAULT;
constexpr double kDefaultFreeFraction = 0.5;

// keeps the calling thread's current device across a library call that has to work on another one
struct DeviceGuard {
  int saved = -1;
  hipError_t enter(int device) {
    if (hipGetDevice(&saved) != hipSuccess) saved = -1;
    return saved == device ? hipSuccess : hipSetDevice(device);
  }
  ~DeviceGuard() {
    int now = -1;
    if (saved >= 0 && hipGetDevice(&now) == hipSuccess && now != saved) (void)hipSetDevice(saved);
  }
};

struct Policy {
  bool spread;
  size_t pool_bytes;
  double free_fraction;
};

Policy resolve(const CaarPlacement* pl) {
  Policy p;
  int policy = pl ? pl->policy : CAAR_PLACE_DEFAULT;
  long long pool = pl ? pl->pool_bytes : 0;
  double frac = pl ? pl->max_free_fraction : 0.0;
  if (policy == CAAR_PLACE_DEFAULT) {  // the caller did not choose: the environment may
    const char* e = std::getenv("CAAR_PLACEMENT");
    policy = (e && std::strcmp(e, "malloc") == 0) ? CAAR_PLACE_MALLOC : CAAR_PLACE_SPREAD;
  }
  if (pool <= 0) {
    const char* e = std::getenv("CAAR_PLACEMENT_POOL_GIB");
    const long g = e ? std::atol(e) : 0;
    pool = g > 0 ? (long long)g << 30 : kDefaultPoolBytes;
  }
  if (!(frac > 0.0)) frac = kDefaultFreeFraction;
  if (frac > 0.9) frac = 0.9;  // never the whole device
  p.spread = policy == CAAR_PLACE_SPREAD;
  p.pool_bytes = (size_t)pool;
  p.free_fraction = frac;
  return p;
}

void release_vmm(CaarArena* a) {
  if (a->va) {
    if (hipDeviceSynchronize() != hipSuccess) ++a->teardown_errors;  // nothing may still be running on this memory
    for (size_t j = 0; j < a->mapped; ++j)                           // one unmap per hipMemMap
      if (hipMemUnmap((char*)a->va + j * kChunk, kChunk) != hipSuccess) {
        (void)hipGetLastError();
        ++a->teardown_errors;
      }
    a->mapped = 0;
  }
  if (a->chunks) {
    for (hipMemGenericAllocationHandle_t h : *a->chunks)
      if (hipMemRelease(h) != hipSuccess) {
        (void)hipGetLastError();
        ++a->teardown_errors;
      }
    delete a->chunks;
    a->chunks = nullptr;
  }
  a->va = nullptr;  // the range stays reserved for the life of the process (see the header of this file)
}

// The VMM route.  Returns false (with everything released) if any step fails.
bool alloc_spread(CaarArena* a, const Policy& pol, const size_t bytes[CAAR_NUM_ARRAYS], double* out[CAAR_NUM_ARRAYS]) {
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
  size_t pool = pol.pool_bytes;
  const size_t cap = (size_t)((double)free_b * pol.free_fraction);
  if (pool > cap) pool = cap;    // leave the rest of the device to the neighbours
  if (pool < total) pool = total;  // (no room to spread: the chunks are taken as they come)
  if (total > free_b) return false;
  const size_t pool_chunks = pool / kChunk, stride = pool_chunks / need;
  std::vector<hipMemGenericAllocationHandle_t> kept;
  kept.reserve(need);
  // create the pool chunk by chunk; keep every stride-th chunk, give the others back at the end (releasing a chunk
  // at once would let the next hipMemCreate return the same memory and the pool would cover nothing)
  std::vector<hipMemGenericAllocationHandle_t> spare;
  spare.reserve(pool_chunks);
  for (size_t c = 0; c < pool_chunks; ++c) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, kChunk, &prop, 0) != hipSuccess) {  // the device filled up (a neighbour allocated meanwhile)
      (void)hipGetLastError();
      break;
    }
    if (c % stride == 0 && kept.size() < need) kept.push_back(h);
    else spare.push_back(h);
  }
  // pool cut short: top up from the spare chunks, evenly over what was created
  if (kept.size() < need && !spare.empty()) {
    const size_t want = need - kept.size(), have = spare.size();
    if (want >= have) {
      kept.insert(kept.end(), spare.begin(), spare.end());
      spare.clear();
    } else {
      std::vector<hipMemGenericAllocationHandle_t> rest;
      rest.reserve(have - want);
      size_t taken = 0;
      for (size_t s = 0; s < have; ++s) {
        if (taken < want && (s * want) / have != ((s + 1) * want) / have) {
          kept.push_back(spare[s]);
          ++taken;
        } else {
          rest.push_back(spare[s]);
        }
      }
      spare.swap(rest);
    }
  }
  for (hipMemGenericAllocationHandle_t h : spare) (void)hipMemRelease(h);
  spare.clear();
  // ... and, with the spare chunks back in the free list, from the device directly
  while (kept.size() < need) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, kChunk, &prop, 0) != hipSuccess) {
      (void)hipGetLastError();
      break;
    }
    kept.push_back(h);
  }
  bool ok = kept.size() == need;
  a->chunks = new (std::nothrow) std::vector<hipMemGenericAllocationHandle_t>();
  if (!a->chunks) ok = false;
  if (!ok) {
    for (hipMemGenericAllocationHandle_t h : kept) (void)hipMemRelease(h);
    delete a->chunks;
    a->chunks = nullptr;
    return false;
  }
  if (hipMemAddressReserve(&a->va, total, kChunk, nullptr, 0) != hipSuccess) {  // chunk-aligned if the driver agrees
    (void)hipGetLastError();
    a->va = nullptr;
    ok = hipMemAddressReserve(&a->va, total, 0, nullptr, 0) == hipSuccess;
  }
  if (ok) {
    a->va_bytes = total;
    // virtual chunk j <- kept chunk (j * 7) mod need (7 is coprime to any need that is not a multiple of 7; if it is, use 11):
    // neighbouring 64 MiB pieces of an array come from distant parts of the pool
    const size_t mul = need % 7 ? 7 : (need % 11 ? 11 : 1);
    a->chunks->resize(need);
    for (size_t j = 0; j < need; ++j) (*a->chunks)[j] = kept[(j * mul) % need];
    for (size_t j = 0; j < need && ok; ++j) {
      ok = hipMemMap((char*)a->va + j * kChunk, kChunk, 0, (*a->chunks)[j], 0) == hipSuccess;
      if (ok) a->mapped = j + 1;
    }
  } else {
    *a->chunks = kept;  // released below
  }
  if (ok) {
    // access for the owning device only: peers (other GPUs over xGMI, RCCL buffers) cannot address a placed arena
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    ok = hipMemSetAccess(a->va, total, &acc, 1) == hipSuccess;
  }
  if (!ok) {
    (void)hipGetLastError();
    release_vmm(a);
    a->teardown_errors = 0;
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

int caar_arrays_alloc_ex(CaarArena** arena, const CaarDims* dims, int device, const CaarPlacement* placement,
                         CaarArrays* out_dev) {
  if (!arena || !dims || !out_dev || dims->num_elems <= 0 || dims->qsize_d < 1 || dims->timelevels < 1 || dims->np < 1 ||
      dims->nlev < 1)
    return CAAR_EINVAL;
  if (placement && (placement->policy < CAAR_PLACE_DEFAULT || placement->policy > CAAR_PLACE_MALLOC ||
                    placement->pool_bytes < 0 || placement->max_free_fraction < 0.0 || placement->max_free_fraction > 1.0))
    return CAAR_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CAAR_ENODEVICE;
  if (device < 0 || device >= ndev) return CAAR_EINVAL;
  DeviceGuard guard;
  hipError_t e = guard.enter(device);
  if (e != hipSuccess) return (int)e;
  CaarArena* a = new (std::nothrow) CaarArena();
  if (!a) return CAAR_ENOMEM;
  a->device = device;
  a->vmm = false;
  a->va = nullptr;
  a->va_bytes = 0;
  a->chunks = nullptr;
  a->mapped = 0;
  a->pool_chunks = a->chunk_bytes = 0;
  a->teardown_errors = 0;
  a->vn0 = nullptr;
  std::memset(a->plain, 0, sizeof(a->plain));
  size_t bytes[CAAR_NUM_ARRAYS], total = 0;
  for (int i = 0; i < CAAR_NUM_ARRAYS; ++i) {
    bytes[i] = sizeof(double) * (size_t)caar_array_len(dims, i);
    total += bytes[i];
  }
  double** out = reinterpret_cast<double**>(out_dev);
  const Policy pol = resolve(placement);
  if (!(pol.spread && total >= kSmall && alloc_spread(a, pol, bytes, out))) {
    for (int i = 0; i < CAAR_NUM_ARRAYS; ++i) {
      e = hipMalloc(&a->plain[i], bytes[i]);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)caar_arrays_free(a);
        return e == hipErrorOutOfMemory ? CAAR_ENOMEM : (int)e;
      }
      out[i] = static_cast<double*>(a->plain[i]);
    }
  }
  a->vn0 = out_dev->elem_derived_vn0;
  *arena = a;
  return CAAR_OK;
}

int caar_arrays_alloc(CaarArena** arena, const CaarDims* dims, int device, CaarArrays* out_dev) {
  return caar_arrays_alloc_ex(arena, dims, device, nullptr, out_dev);
}

int caar_arrays_free(CaarArena* a) {
  if (!a) return CAAR_EINVAL;
  DeviceGuard guard;  // may run from a finaliser on whatever thread and device happen to be current
  (void)guard.enter(a->device);
  caar::window_tuner_forget(a->vn0, a->device);  // a later set at the same address starts from the default policy
  release_vmm(a);
  for (int i = 0; i < CAAR_NUM_ARRAYS; ++i)
    if (a->plain[i] && hipFree(a->plain[i]) != hipSuccess) {
      (void)hipGetLastError();
      ++a->teardown_errors;
    }
  const int bad = a->teardown_errors;
  delete a;
  return bad ? (int)hipErrorUnknown : CAAR_OK;
}

int caar_arrays_placement(const CaarArena* a, long long* pool_chunks, long long* chunk_bytes) {
  if (!a) return CAAR_EINVAL;
  if (pool_chunks) *pool_chunks = a->pool_chunks;
  if (chunk_bytes) *chunk_bytes = a->chunk_bytes;
  return a->vmm ? 1 : 0;
}

}  // extern "C"
