// caar_membench.hip — measurement utilities for the roofline (not on the product path).
//
//  * stream_copy: plain device copy, 8 or 16 bytes per lane, grid-stride — the
//    measured HBM ceiling quoted next to the 8 TB/s spec peak in bench.py/DESIGN.md and
//    the calibration run for the FETCH_SIZE/WRITE_SIZE counters
//    (/opt/skills/guides/MI355X_MICROARCH.md, HBM section: calibrate on a known byte count
//    in your own access pattern).
//  * traffic_skeleton: touches exactly the bytes compute_and_apply_rhs touches for NP=4
//    (same arrays, same tile/lane addressing, same load/store widths) with no
//    arithmetic worth mentioning: the bandwidth this access pattern can reach when
//    nothing but memory limits it.
#include <hip/hip_runtime.h>

#include "caar_kernel_args.h"

namespace caar {

template <typename V>
__global__ void stream_copy_kernel(V* __restrict__ dst, const V* __restrict__ src, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i];
}

hipError_t launch_stream_copy(double* dst, const double* src, size_t n_doubles, int lane_bytes,
                              hipStream_t stream) {
  const int threads = 256, blocks = 256 * 8;
  if (lane_bytes == 16 && n_doubles % 2 == 0)
    hipLaunchKernelGGL(stream_copy_kernel<double2>, dim3(blocks), dim3(threads), 0, stream,
                       reinterpret_cast<double2*>(dst), reinterpret_cast<const double2*>(src), n_doubles / 2);
  else
    hipLaunchKernelGGL(stream_copy_kernel<double>, dim3(blocks), dim3(threads), 0, stream, dst, src, n_doubles);
  return hipGetLastError();
}

// Tuned device copy: the ceiling to quote next to the 8 TB/s spec peak (MI355X_MICROARCH.md: a float4
// copy reaches 6.29 TB/s = 79 %).  16 bytes per lane, UNROLL independent loads in flight per lane
// before the first store, a workgroup owns contiguous chunks of 256*UNROLL*16 bytes (each wave
// instruction is one 1 KiB segment), grid = CUs x WG_PER_CU resident workgroups walking the buffer
// chunk-strided, optional non-temporal policy.  The plain grid-stride kernel above (one 8-byte
// access in flight per lane and iteration) stays as the calibration run of the PMC counters.
typedef double v2d __attribute__((ext_vector_type(2)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void stream_copy_tuned_kernel(v2d* __restrict__ dst, const v2d* __restrict__ src,
                                                                size_t n16) {
  constexpr size_t CHUNK = 256 * UNROLL;
  const size_t nchunks = n16 / CHUNK;
  for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const v2d* s = src + c * CHUNK + threadIdx.x;
    v2d* d = dst + c * CHUNK + threadIdx.x;
    v2d x[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) x[u] = NT ? __builtin_nontemporal_load(s + u * 256) : s[u * 256];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (NT) __builtin_nontemporal_store(x[u], d + u * 256);
      else d[u * 256] = x[u];
    }
  }
  // tail (less than one chunk): first workgroup, one access per lane and iteration
  if (blockIdx.x == 0)
    for (size_t i = nchunks * CHUNK + threadIdx.x; i < n16; i += 256) dst[i] = src[i];
}

struct CopyVariant {
  const char* what;
  int unroll, nt, wg_per_cu;
};
static const CopyVariant kCopyVariants[] = {
    {"16 B/lane, 4 loads in flight, nt, 8 workgroups/CU", 4, 1, 8},
    {"16 B/lane, 8 loads in flight, nt, 8 workgroups/CU", 8, 1, 8},
    {"16 B/lane, 4 loads in flight, nt, 4 workgroups/CU", 4, 1, 4},
    {"16 B/lane, 8 loads in flight, nt, 4 workgroups/CU", 8, 1, 4},
    {"16 B/lane, 4 loads in flight, default policy, 8 workgroups/CU", 4, 0, 8},
    {"16 B/lane, 8 loads in flight, default policy, 8 workgroups/CU", 8, 0, 8},
    {"16 B/lane, 4 loads in flight, default policy, 4 workgroups/CU", 4, 0, 4},
    {"16 B/lane, 2 loads in flight, nt, 8 workgroups/CU", 2, 1, 8},
    {"16 B/lane, 2 loads in flight, default policy, 8 workgroups/CU", 2, 0, 8},
    {"16 B/lane, 4 loads in flight, nt, 16 workgroups/CU (grid oversubscribed 2x)", 4, 1, 16},
    {"16 B/lane, 4 loads in flight, nt, 32 workgroups/CU (grid oversubscribed 4x)", 4, 1, 32},
    {"16 B/lane, 4 loads in flight, nt, one 16 KiB chunk per workgroup", 4, 1, 0},
    {"16 B/lane, 2 loads in flight, nt, one 8 KiB chunk per workgroup", 2, 1, 0},
    {"16 B/lane, 2 loads in flight, nt, 32 workgroups/CU", 2, 1, 32},
    {"16 B/lane, 4 loads in flight, default policy, one 16 KiB chunk per workgroup", 4, 0, 0},
    {"16 B/lane, 1 load in flight, nt, one 4 KiB chunk per workgroup", 1, 1, 0},
};
int stream_copy_tuned_variants() { return (int)(sizeof(kCopyVariants) / sizeof(kCopyVariants[0])); }
const char* stream_copy_tuned_info(int v) {
  return (v >= 0 && v < stream_copy_tuned_variants()) ? kCopyVariants[v].what : nullptr;
}

hipError_t launch_stream_copy_tuned(double* dst, const double* src, size_t n_doubles, int variant, hipStream_t stream) {
  if (variant < 0 || variant >= stream_copy_tuned_variants() || (n_doubles & 1)) return hipErrorInvalidValue;
  const CopyVariant& cv = kCopyVariants[variant];
  int dev = 0, cus = 256;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
    cus = p.multiProcessorCount;
  const size_t n16 = n_doubles / 2;
  size_t nblocks = (size_t)cus * cv.wg_per_cu;
  if (cv.wg_per_cu == 0) {  // one chunk per workgroup: the hardware dispatcher walks the buffer
    nblocks = n16 / (256 * (size_t)cv.unroll);
    if (nblocks == 0) nblocks = 1;
    if (nblocks > 0x7fffffff) return hipErrorInvalidValue;
  }
  const dim3 grid((unsigned)nblocks), block(256);
  v2d* d = reinterpret_cast<v2d*>(dst);
  const v2d* s = reinterpret_cast<const v2d*>(src);
#define CAAR_COPY(U, N) hipLaunchKernelGGL((stream_copy_tuned_kernel<U, N>), grid, block, 0, stream, d, s, n16)
  if (cv.unroll == 1) { if (cv.nt) CAAR_COPY(1, true); else CAAR_COPY(1, false); }
  else if (cv.unroll == 2) { if (cv.nt) CAAR_COPY(2, true); else CAAR_COPY(2, false); }
  else if (cv.unroll == 4) { if (cv.nt) CAAR_COPY(4, true); else CAAR_COPY(4, false); }
  else { if (cv.nt) CAAR_COPY(8, true); else CAAR_COPY(8, false); }
#undef CAAR_COPY
  return hipGetLastError();
}

// cache-policy codes: 0 default, 1 nt (non-temporal), 2 sc1 (agent-scope: bypasses the L1 on
// loads, drops the L2 line on stores; MI355X_MICROARCH.md "stores of each flavour")
template <int POL, typename T>
__device__ __forceinline__ T ld(const T* p) {
  if constexpr (POL == 1) return __builtin_nontemporal_load(p);
  else if constexpr (POL == 2) {
    if constexpr (sizeof(T) == 8) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else {
      T r;
      const double* q = reinterpret_cast<const double*>(p);
      r.x = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      r.y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return r;
    }
  } else return *p;
}
template <int POL, typename T>
__device__ __forceinline__ void st(T* p, T v) {
  if constexpr (POL == 1) __builtin_nontemporal_store(v, p);
  else if constexpr (POL == 2) {
    if constexpr (sizeof(T) == 8) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else {
      double* q = reinterpret_cast<double*>(p);
      __hip_atomic_store(q, (double)v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(q + 1, (double)v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else *p = v;
}
__device__ __forceinline__ double2 ld2(const double2* p, bool) { return *p; }

// NTL / NTS: cache policy of the loads / stores (0 default, 1 nt, 2 sc1); ALL_FIRST: issue every load of the element
// before the first store (maximum bytes in flight) instead of tile by tile.
// PP = 16: NP=4 (a 64-lane tile = 4 levels); PP = 64: NP=8 (a tile = one level).
// WGW > 0: the element's waves are spread over workgroups of WGW waves (the skeleton has no cross-level
// dependency): many small, short-lived workgroups per CU instead of one fat one — probes what the launch
// shape alone costs.
// ORDER (with ALL_FIRST): 0 = tile-major (a tile's 7 output arrays, then the next tile: 8 streams x 512 B each time);
// 1 = stores array-major (one array's TPW tiles back to back = TPW x 512 B contiguous per wave and array: 3 KiB at
// TPW = 6, 6 KiB for the (u, v) pairs), 2 = loads array-major as well.  Probes whether DRAM prefers longer bursts per
// stream (round 2: write streams are what costs, profiles/r02/stream_probe_rw.log).
// ANTL / ANTS: the policy of the three read-modify-write accumulators (vn0, omega_p, eta_dot_dpdn), which the hybrid
// form (traffic_skeleton_np4_hybrid) switches per element like the real kernel's hybrid cache policy does.
template <int NLEV, int TPW, int NTL, int NTS, bool ALL_FIRST, int PP, int WGW, int ORDER, int ANTL, int ANTS>
__device__ __forceinline__ void traffic_skeleton_body(const KernelArgs& k) {
  static_assert(ORDER == 0 || (ANTL == NTL && ANTS == NTS), "array-major orders: one policy");
  constexpr int BLK = NLEV * PP;
  constexpr int WAVES = NLEV * PP / 64 / TPW;            // waves per element
  constexpr int WG_PER_ELEM = WGW ? WAVES / WGW : 1;
  static_assert(!WGW || WAVES % WGW == 0, "wave split");
  const int tid = threadIdx.x, lane = tid & 63, pt = lane & (PP - 1);
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6) + (WGW ? (int)(blockIdx.x % WG_PER_ELEM) * WGW : 0);
  const unsigned ulane = lane;
  const long long ie_s = element_of_block(k, WGW ? blockIdx.x / WG_PER_ELEM : blockIdx.x);
  if (ie_s < 0) return;
  const size_t ie = (size_t)ie_s, tl = (size_t)k.timelevels;
  const size_t wb = (size_t)w * (TPW * 64);
  const double* dp_n0 = k.dp3d + (ie * tl + k.n0) * BLK + wb;
  const double* vv_n0 = k.v + ((ie * tl + k.n0) * BLK + wb) * 2;
  const double* T_n0 = k.T + (ie * tl + k.n0) * BLK + wb;
  const double* Qdp = k.Qdp + ((ie * k.qsize_d + 0) * 2 + (k.qn0 >= 0 ? k.qn0 : 0)) * BLK + wb;
  const double* vv_nm1 = k.v + ((ie * tl + k.nm1) * BLK + wb) * 2;
  const double* T_nm1 = k.T + (ie * tl + k.nm1) * BLK + wb;
  const double* dp_nm1 = k.dp3d + (ie * tl + k.nm1) * BLK + wb;
  double* vv_np1 = k.v + ((ie * tl + k.np1) * BLK + wb) * 2;
  double* T_np1 = k.T + (ie * tl + k.np1) * BLK + wb;
  double* dp_np1 = k.dp3d + (ie * tl + k.np1) * BLK + wb;
  double* vvn0 = k.vn0 + (ie * BLK + wb) * 2;
  double* omega_p = k.omega_p + ie * BLK + wb;
  double* phi = k.phi + ie * BLK + wb;
  const double* pecnd = k.pecnd + ie * BLK + wb;
  double* eta = k.eta_dot_dpdn + ie * (BLK + PP) + wb;
  double* eta_last = k.eta_dot_dpdn + ie * (BLK + PP) + BLK;
  typedef double v2 __attribute__((ext_vector_type(2)));
  // 13 metric values per point, read by every lane like the real kernel's LDS staging source
  double g = k.fcor[ie * PP + pt] + k.spheremp[ie * PP + pt] + k.metdet[ie * PP + pt] +
             k.rmetdet[ie * PP + pt] + k.phis[ie * PP + pt];
#pragma unroll
  for (int j = 0; j < 4; ++j) g += k.D[(ie * PP + pt) * 4 + j] + k.Dinv[(ie * PP + pt) * 4 + j];
  g *= 0.0;
  double a[TPW], tn[TPW], dn[TPW], om[TPW], et[TPW];
  v2 uv[TPW], um[TPW], un[TPW];
  if constexpr (ALL_FIRST && ORDER >= 1) {
    if constexpr (ORDER == 2) {
#define CAAR_SK_LD(dst, expr)                          \
  _Pragma("unroll") for (int r = 0; r < TPW; ++r) {    \
    const unsigned off = r * 64 + ulane;               \
    dst[r] = expr;                                     \
  }
      double b0[TPW], b1[TPW], b2[TPW], b3[TPW];
      CAAR_SK_LD(b0, ld<NTL>(dp_n0 + off))
      CAAR_SK_LD(uv, ld<NTL>(reinterpret_cast<const v2*>(vv_n0) + off))
      CAAR_SK_LD(b1, ld<NTL>(T_n0 + off))
      CAAR_SK_LD(b2, ld<NTL>(Qdp + off))
      CAAR_SK_LD(um, ld<NTL>(reinterpret_cast<const v2*>(vv_nm1) + off))
      CAAR_SK_LD(tn, ld<NTL>(T_nm1 + off))
      CAAR_SK_LD(dn, ld<NTL>(dp_nm1 + off))
      CAAR_SK_LD(un, ld<NTL>(reinterpret_cast<const v2*>(vvn0) + off))
      CAAR_SK_LD(om, ld<NTL>(omega_p + off))
      CAAR_SK_LD(b3, ld<NTL>(pecnd + off))
      CAAR_SK_LD(et, ld<NTL>(eta + off))
#undef CAAR_SK_LD
#pragma unroll
      for (int r = 0; r < TPW; ++r) a[r] = b0[r] + b1[r] + b2[r] + b3[r] + g;
    } else {
#pragma unroll
      for (int r = 0; r < TPW; ++r) {
        const unsigned off = r * 64 + ulane;
        a[r] = ld<NTL>(dp_n0 + off) + ld<NTL>(T_n0 + off) + ld<NTL>(Qdp + off) + ld<NTL>(pecnd + off) + g;
        uv[r] = ld<NTL>(reinterpret_cast<const v2*>(vv_n0) + off);
        um[r] = ld<NTL>(reinterpret_cast<const v2*>(vv_nm1) + off);
        un[r] = ld<NTL>(reinterpret_cast<const v2*>(vvn0) + off);
        tn[r] = ld<NTL>(T_nm1 + off);
        dn[r] = ld<NTL>(dp_nm1 + off);
        om[r] = ld<NTL>(omega_p + off);
        et[r] = ld<NTL>(eta + off);
      }
    }
#define CAAR_SK_ST(ptr, expr)                          \
  _Pragma("unroll") for (int r = 0; r < TPW; ++r) {    \
    const unsigned off = r * 64 + ulane;               \
    st<NTS>(ptr + off, expr);                          \
  }
    CAAR_SK_ST(reinterpret_cast<v2*>(vv_np1), uv[r] + um[r])
    CAAR_SK_ST(T_np1, tn[r] + a[r])
    CAAR_SK_ST(dp_np1, dn[r] + a[r])
    CAAR_SK_ST(phi, a[r])
    CAAR_SK_ST(omega_p, om[r] + a[r] * 0.0)
    CAAR_SK_ST(reinterpret_cast<v2*>(vvn0), un[r])
    CAAR_SK_ST(eta, et[r] + 0.0)
#undef CAAR_SK_ST
    if (w == 0 && lane < PP) eta_last[ulane] = eta_last[ulane] + 0.0;
    return;
  }
#pragma unroll
  for (int r = 0; r < TPW; ++r) {
    const unsigned off = r * 64 + ulane;
    a[r] = ld<NTL>(dp_n0 + off) + ld<NTL>(T_n0 + off) + ld<NTL>(Qdp + off) + ld<NTL>(pecnd + off) + g;
    uv[r] = ld<NTL>(reinterpret_cast<const v2*>(vv_n0) + off);
    um[r] = ld<NTL>(reinterpret_cast<const v2*>(vv_nm1) + off);
    un[r] = ld<ANTL>(reinterpret_cast<const v2*>(vvn0) + off);
    tn[r] = ld<NTL>(T_nm1 + off);
    dn[r] = ld<NTL>(dp_nm1 + off);
    om[r] = ld<ANTL>(omega_p + off);
    et[r] = ld<ANTL>(eta + off);
    if (!ALL_FIRST) {
      st<NTS>(reinterpret_cast<v2*>(vv_np1) + off, uv[r] + um[r]);
      st<NTS>(T_np1 + off, tn[r] + a[r]);
      st<NTS>(dp_np1 + off, dn[r] + a[r]);
      st<NTS>(phi + off, a[r]);
      st<ANTS>(omega_p + off, om[r] + a[r] * 0.0);
      st<ANTS>(reinterpret_cast<v2*>(vvn0) + off, un[r]);
      st<ANTS>(eta + off, et[r] + 0.0);
    }
  }
  if (ALL_FIRST) {
#pragma unroll
    for (int r = 0; r < TPW; ++r) {
      const unsigned off = r * 64 + ulane;
      st<NTS>(reinterpret_cast<v2*>(vv_np1) + off, uv[r] + um[r]);
      st<NTS>(T_np1 + off, tn[r] + a[r]);
      st<NTS>(dp_np1 + off, dn[r] + a[r]);
      st<NTS>(phi + off, a[r]);
      st<ANTS>(omega_p + off, om[r] + a[r] * 0.0);
      st<ANTS>(reinterpret_cast<v2*>(vvn0) + off, un[r]);
      st<ANTS>(eta + off, et[r] + 0.0);
    }
  }
  if (w == 0 && lane < PP) eta_last[ulane] = eta_last[ulane] + 0.0;
}

template <int NLEV, int TPW, int NTL, int NTS, bool ALL_FIRST, int PP = 16, int WGW = 0, int ORDER = 0>
__global__ __launch_bounds__((WGW ? WGW : NLEV * PP / 64 / TPW) * 64) void traffic_skeleton_np4(const KernelArgs k) {
  traffic_skeleton_body<NLEV, TPW, NTL, NTS, ALL_FIRST, PP, WGW, ORDER, NTL, NTS>(k);
}
// The hybrid cache policy's traffic without its arithmetic: everything non-temporal except the accumulators of the
// k.cache_count elements element_is_cached picks, which use the default policy (what the default NP=4 kernels do).
// MINW: the register budget of the real kernel's launch shape, so that as many workgroups share a CU as there (2).
// WGW > 0: the element's waves spread over workgroups of WGW waves (as in traffic_skeleton_np4); AF: all loads first.
template <int NLEV, int TPW, int MINW, int WGW = 0, bool AF = true>
__global__ __launch_bounds__((WGW ? WGW : NLEV * 16 / 64 / TPW) * 64, MINW) void traffic_skeleton_np4_hybrid(const KernelArgs k) {
  constexpr int WG_PER_ELEM = WGW ? NLEV * 16 / 64 / TPW / WGW : 1;
  const long long ie_s = element_of_block(k, WGW ? blockIdx.x / WG_PER_ELEM : blockIdx.x);
  if (ie_s < 0) return;
  if (element_is_cached(k, ie_s)) traffic_skeleton_body<NLEV, TPW, 1, 1, AF, 16, WGW, 0, 0, 0>(k);
  else traffic_skeleton_body<NLEV, TPW, 1, 1, AF, 16, WGW, 0, 1, 1>(k);
}

// Same bytes as traffic_skeleton_np4<NLEV, 2, ...> but every access 16 bytes per lane: the two
// tiles of a wave are contiguous (128 doubles of a scalar field = one dwordx4 per lane; 256
// doubles of v = two).  Probes whether wider accesses alone raise the ceiling.
template <int NLEV, int POL>
__global__ __launch_bounds__(NLEV / 8 * 64) void traffic_skeleton_np4_w16(const KernelArgs k) {
  constexpr int PP = 16, BLK = NLEV * PP;
  typedef double v2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63, pt = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned ulane = lane;
  const long long ie_s = element_of_block(k, blockIdx.x);
  if (ie_s < 0) return;
  const size_t ie = (size_t)ie_s, tl = (size_t)k.timelevels, wb = (size_t)w * 128;
  auto S = [&](const double* base) { return reinterpret_cast<const v2*>(base + wb); };
  auto SW = [&](double* base) { return reinterpret_cast<v2*>(base + wb); };
  const v2* dp_n0 = S(k.dp3d + (ie * tl + k.n0) * BLK);
  const v2* T_n0 = S(k.T + (ie * tl + k.n0) * BLK);
  const v2* Qdp = S(k.Qdp + ((ie * k.qsize_d + 0) * 2 + (k.qn0 >= 0 ? k.qn0 : 0)) * BLK);
  const v2* T_nm1 = S(k.T + (ie * tl + k.nm1) * BLK);
  const v2* dp_nm1 = S(k.dp3d + (ie * tl + k.nm1) * BLK);
  const v2* pecnd = S(k.pecnd + ie * BLK);
  v2* omega_p = SW(k.omega_p + ie * BLK);
  v2* phi = SW(k.phi + ie * BLK);
  v2* eta = SW(k.eta_dot_dpdn + ie * (BLK + PP));
  v2* T_np1 = SW(k.T + (ie * tl + k.np1) * BLK);
  v2* dp_np1 = SW(k.dp3d + (ie * tl + k.np1) * BLK);
  const v2* v_n0 = reinterpret_cast<const v2*>(k.v + ((ie * tl + k.n0) * BLK + wb) * 2);
  const v2* v_nm1 = reinterpret_cast<const v2*>(k.v + ((ie * tl + k.nm1) * BLK + wb) * 2);
  v2* v_np1 = reinterpret_cast<v2*>(k.v + ((ie * tl + k.np1) * BLK + wb) * 2);
  v2* vn0 = reinterpret_cast<v2*>(k.vn0 + (ie * BLK + wb) * 2);
  double* eta_last = k.eta_dot_dpdn + ie * (BLK + PP) + BLK;
  double g = k.fcor[ie * PP + pt] + k.spheremp[ie * PP + pt] + k.metdet[ie * PP + pt] +
             k.rmetdet[ie * PP + pt] + k.phis[ie * PP + pt];
#pragma unroll
  for (int j = 0; j < 4; ++j) g += k.D[(ie * PP + pt) * 4 + j] + k.Dinv[(ie * PP + pt) * 4 + j];
  g *= 0.0;
  const v2 a = ld<POL>(dp_n0 + ulane) + ld<POL>(T_n0 + ulane) + ld<POL>(Qdp + ulane) + ld<POL>(pecnd + ulane) + g;
  const v2 uv0 = ld<POL>(v_n0 + ulane), uv1 = ld<POL>(v_n0 + 64 + ulane);
  const v2 um0 = ld<POL>(v_nm1 + ulane), um1 = ld<POL>(v_nm1 + 64 + ulane);
  const v2 un0 = ld<POL>(vn0 + ulane), un1 = ld<POL>(vn0 + 64 + ulane);
  const v2 tn = ld<POL>(T_nm1 + ulane), dn = ld<POL>(dp_nm1 + ulane), om = ld<POL>(omega_p + ulane),
           et = ld<POL>(eta + ulane);
  st<POL>(v_np1 + ulane, uv0 + um0);
  st<POL>(v_np1 + 64 + ulane, uv1 + um1);
  st<POL>(T_np1 + ulane, tn + a);
  st<POL>(dp_np1 + ulane, dn + a);
  st<POL>(phi + ulane, a);
  st<POL>(omega_p + ulane, om + a * 0.0);
  st<POL>(vn0 + ulane, un0);
  st<POL>(vn0 + 64 + ulane, un1);
  st<POL>(eta + ulane, et + 0.0);
  if (tid < PP) eta_last[ulane] = eta_last[ulane] + 0.0;
}

template <int NLEV, int TPW, int NTL, int NTS, bool AF>
static void skel(const KernelArgs& k, int n, hipStream_t s) {
  hipLaunchKernelGGL((traffic_skeleton_np4<NLEV, TPW, NTL, NTS, AF>), dim3(k.per_xcd ? 8 * k.per_xcd : n), dim3(NLEV / 4 / TPW * 64), 0, s, k);
}

hipError_t launch_traffic_skeleton(const KernelArgs& k, int nlev, int variant, int num_elems, hipStream_t s) {
  if (nlev == 72) {
    switch (variant) {
      case 0: skel<72, 3, 0, 0, false>(k, num_elems, s); break;
      case 1: skel<72, 2, 0, 0, false>(k, num_elems, s); break;
      case 2: skel<72, 6, 0, 0, false>(k, num_elems, s); break;
      case 3: skel<72, 3, 1, 1, false>(k, num_elems, s); break;
      case 4: skel<72, 3, 0, 1, false>(k, num_elems, s); break;
      case 5: skel<72, 3, 1, 0, false>(k, num_elems, s); break;
      case 6: skel<72, 3, 0, 0, true>(k, num_elems, s); break;
      case 7: skel<72, 2, 0, 0, true>(k, num_elems, s); break;
      case 8: skel<72, 2, 1, 1, true>(k, num_elems, s); break;
      case 9: skel<72, 9, 0, 0, true>(k, num_elems, s); break;
      case 10: skel<72, 2, 2, 1, true>(k, num_elems, s); break;
      case 11: skel<72, 2, 1, 2, true>(k, num_elems, s); break;
      case 12: skel<72, 2, 2, 2, true>(k, num_elems, s); break;
      // small workgroups: <TPW, WGW> -> (18 / TPW / WGW) workgroups of WGW waves per element
      case 15: hipLaunchKernelGGL((traffic_skeleton_np4<72, 1, 1, 1, true, 16, 1>), dim3(num_elems * 18), dim3(64), 0, s, k); break;
      case 16: hipLaunchKernelGGL((traffic_skeleton_np4<72, 2, 1, 1, true, 16, 1>), dim3(num_elems * 9), dim3(64), 0, s, k); break;
      case 17: hipLaunchKernelGGL((traffic_skeleton_np4<72, 1, 1, 1, true, 16, 2>), dim3(num_elems * 9), dim3(128), 0, s, k); break;
      case 18: hipLaunchKernelGGL((traffic_skeleton_np4<72, 1, 1, 1, true, 16, 6>), dim3(num_elems * 3), dim3(384), 0, s, k); break;
      case 19: hipLaunchKernelGGL((traffic_skeleton_np4<72, 2, 1, 1, true, 16, 3>), dim3(num_elems * 3), dim3(192), 0, s, k); break;
      case 20: hipLaunchKernelGGL((traffic_skeleton_np4<72, 1, 1, 1, false, 16, 1>), dim3(num_elems * 18), dim3(64), 0, s, k); break;
      case 22: skel<72, 6, 1, 1, true>(k, num_elems, s); break;   // the launch shape of the default kernel of rounds 2-3 (3 waves x 6 tiles), nt
      // the same shape with each output array's stores issued as one 3 KiB (u, v: 6 KiB) burst per wave (23) and the loads
      // array-major too (24); 25/26: the same two at 2 waves x 9 tiles (4.5 / 9 KiB bursts)
      case 23: hipLaunchKernelGGL((traffic_skeleton_np4<72, 6, 1, 1, true, 16, 0, 1>), dim3(k.per_xcd ? 8 * k.per_xcd : num_elems), dim3(192), 0, s, k); break;
      case 24: hipLaunchKernelGGL((traffic_skeleton_np4<72, 6, 1, 1, true, 16, 0, 2>), dim3(k.per_xcd ? 8 * k.per_xcd : num_elems), dim3(192), 0, s, k); break;
      case 25: hipLaunchKernelGGL((traffic_skeleton_np4<72, 9, 1, 1, true, 16, 0, 1>), dim3(k.per_xcd ? 8 * k.per_xcd : num_elems), dim3(128), 0, s, k); break;
      case 26: hipLaunchKernelGGL((traffic_skeleton_np4<72, 9, 1, 1, true, 16, 0, 2>), dim3(k.per_xcd ? 8 * k.per_xcd : num_elems), dim3(128), 0, s, k); break;
      case 27: skel<72, 9, 1, 1, true>(k, num_elems, s); break;   // 2 waves x 9 tiles, tile-major (the comparator of 25/26)
      // the HYBRID policy's traffic (round 4): the default kernel's cache policy and window, no arithmetic; 3 waves x 6 tiles
      // (28) and 6 waves x 3 tiles (29) — the skeleton has no uneven 4-wave shape; both hold all loads of an element in flight
      case 28: hipLaunchKernelGGL((traffic_skeleton_np4_hybrid<72, 6, 2>), dim3(k.per_xcd ? 8 * k.per_xcd : num_elems), dim3(192), 0, s, k); break;
      case 29: hipLaunchKernelGGL((traffic_skeleton_np4_hybrid<72, 3, 1>), dim3(k.per_xcd ? 8 * k.per_xcd : num_elems), dim3(384), 0, s, k); break;
      // hybrid, other shapes (plain element mapping): 18 / 9 one-wave workgroups per element, 9 two-wave ones, 3 x 6 tile by tile, 2 x 9, 9 x 2
      case 30: hipLaunchKernelGGL((traffic_skeleton_np4_hybrid<72, 1, 1, 1>), dim3(num_elems * 18), dim3(64), 0, s, k); break;
      case 31: hipLaunchKernelGGL((traffic_skeleton_np4_hybrid<72, 2, 1, 1>), dim3(num_elems * 9), dim3(64), 0, s, k); break;
      case 32: hipLaunchKernelGGL((traffic_skeleton_np4_hybrid<72, 1, 1, 2>), dim3(num_elems * 9), dim3(128), 0, s, k); break;
      case 33: hipLaunchKernelGGL((traffic_skeleton_np4_hybrid<72, 6, 2, 0, false>), dim3(k.per_xcd ? 8 * k.per_xcd : num_elems), dim3(192), 0, s, k); break;
      case 34: hipLaunchKernelGGL((traffic_skeleton_np4_hybrid<72, 9, 1>), dim3(k.per_xcd ? 8 * k.per_xcd : num_elems), dim3(128), 0, s, k); break;
      case 35: hipLaunchKernelGGL((traffic_skeleton_np4_hybrid<72, 2, 1>), dim3(k.per_xcd ? 8 * k.per_xcd : num_elems), dim3(576), 0, s, k); break;
      case 13: hipLaunchKernelGGL((traffic_skeleton_np4_w16<72, 1>), dim3(k.per_xcd ? 8 * k.per_xcd : num_elems), dim3(576), 0, s, k); break;
      case 14: hipLaunchKernelGGL((traffic_skeleton_np4_w16<72, 0>), dim3(k.per_xcd ? 8 * k.per_xcd : num_elems), dim3(576), 0, s, k); break;
      default: return hipErrorInvalidValue;
    }
  } else if (nlev == 128) {
    switch (variant) {
      case 0: skel<128, 4, 0, 0, false>(k, num_elems, s); break;
      case 1: skel<128, 2, 0, 0, false>(k, num_elems, s); break;
      case 2: skel<128, 4, 1, 1, false>(k, num_elems, s); break;
      case 3: skel<128, 4, 0, 0, true>(k, num_elems, s); break;
      // the HYBRID policy's traffic (round 4), NLEV=128: 4 waves x 8 tiles (the default kernel's shape) and 8 x 4
      case 28: hipLaunchKernelGGL((traffic_skeleton_np4_hybrid<128, 8, 2>), dim3(k.per_xcd ? 8 * k.per_xcd : num_elems), dim3(256), 0, s, k); break;
      case 29: hipLaunchKernelGGL((traffic_skeleton_np4_hybrid<128, 4, 1>), dim3(k.per_xcd ? 8 * k.per_xcd : num_elems), dim3(512), 0, s, k); break;
      default: return hipErrorInvalidValue;
    }
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// NP=8 NLEV=72: the launch shape of caar_np8_kernel (8 waves x 9 levels)
hipError_t launch_traffic_skeleton_np8(const KernelArgs& k, int nlev, int variant, int num_elems, hipStream_t s) {
  if (nlev != 72) return hipErrorInvalidValue;
  const dim3 grid(k.per_xcd ? 8 * k.per_xcd : num_elems), block(512);
  switch (variant) {
    case 0: hipLaunchKernelGGL((traffic_skeleton_np4<72, 9, 1, 1, false, 64>), grid, block, 0, s, k); break;
    case 1: hipLaunchKernelGGL((traffic_skeleton_np4<72, 9, 0, 0, false, 64>), grid, block, 0, s, k); break;
    case 2: hipLaunchKernelGGL((traffic_skeleton_np4<72, 9, 1, 1, true, 64>), grid, block, 0, s, k); break;
    // the element's 8 waves spread over workgroups of 4 / 2 / 1 waves: what more workgroups per CU would be worth
    case 3: hipLaunchKernelGGL((traffic_skeleton_np4<72, 9, 1, 1, false, 64, 4>), dim3(num_elems * 2), dim3(256), 0, s, k); break;
    case 4: hipLaunchKernelGGL((traffic_skeleton_np4<72, 9, 1, 1, false, 64, 2>), dim3(num_elems * 4), dim3(128), 0, s, k); break;
    case 5: hipLaunchKernelGGL((traffic_skeleton_np4<72, 9, 1, 1, false, 64, 1>), dim3(num_elems * 8), dim3(64), 0, s, k); break;
    case 6: hipLaunchKernelGGL((traffic_skeleton_np4<72, 3, 1, 1, false, 64, 4>), dim3(num_elems * 6), dim3(256), 0, s, k); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace caar
