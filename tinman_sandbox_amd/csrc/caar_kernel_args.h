// caar_kernel_args.h — kernel argument block shared by the HIP kernels and the ABI layer.
#pragma once

#include <hip/hip_runtime.h>

// CAAR_EXTRA_NLEV = 1 (build option, default 0): besides the BASELINE configurations (NP=4 NLEV 72 / 128, NP=8 NLEV 72) and
// the run-time-level-count kernel that serves every other NLEV in 2..256, also compile
//   * launch shapes specialised for NLEV 26, 30, 32, 60, 64, 80, 96 (84.7-87 % of peak against 81-87 % through the run-time
//     count, profiles/r03/anylev_bench.log) and the step loops of NLEV 80 / 64 / 60,
//   * the Eulerian (rsplit == 0) form of the run-time-level kernel beyond 128 levels (8 waves x 8 tiles: 65-90 spilled VGPRs).
// None of these is a SURVEY section 8 row; round 3 built them, round 4 took them out of the default library (126 -> 64 NP=4
// kernel instantiations; every kernel the default build can reach is free of register spills except the Eulerian form at
// NLEV=128, whose two-workgroup shape spills 14 VGPRs and is still the fastest measured).
#ifndef CAAR_EXTRA_NLEV
#define CAAR_EXTRA_NLEV 0
#endif

namespace caar {

// Device pointers to element 0 of each array (layouts: include/caar.h) plus the
// scalars of Homme::Control / Constants / HVCoord (data_structures.hpp:10-76).
struct KernelArgs {
  const double* D;
  const double* Dinv;
  const double* fcor;
  const double* spheremp;
  const double* metdet;
  const double* rmetdet;
  double* dp3d;
  double* v;
  double* T;
  const double* phis;
  const double* Qdp;
  double* eta_dot_dpdn;
  double* omega_p;
  double* phi;
  const double* pecnd;
  double* vn0;
  const double* Dvv;  // np*np, row-major Dvv[i][j]
  const double* hybi; // nlev+1 interface coefficients, read only when vadv != 0
  int vadv;           // 1: rsplit == 0, Eulerian vertical coordinate (eta_dot_dpdn, vertical advection)
  int nets;           // first element of this launch
  int nelem;          // elements in this launch
  int per_xcd;        // 0: element = nets + blockIdx.x; else XCD-chunked mapping (element_of_block)
  // hybrid cache policy: cache_count of the arrays' cache_n elements, spread evenly over the WHOLE element range of the arrays
  // (not over the launch), keep their accumulators in the memory-side cache (0: none) — launches on sub-ranges, one after the
  // other or side by side on several streams, together keep exactly the set one launch over everything keeps.  The kernels
  // get the ratio: floor(cache_count * 2^32 / cache_n), formed on the host (element_is_cached)
  unsigned cache_q_lo, cache_q_hi;
  int n0, np1, nm1;
  int qn0;            // -1: dry
  int qsize_d, timelevels;
  int nlev;           // vertical levels (read by the kernels compiled for a run-time level count)
  double dt2;
  double rrearth;
  double eta_ave_w;
  double rv_over_rd_m1;  // Rwater_vapor/Rgas - 1.0 (P:151), formed on the host in fp64
  double Rgas;
  double kappa;
  double p_top;          // hyai[0]*ps0 (P:84)
};

// Workgroup -> element.  Workgroups are dealt round-robin over the 8 XCDs
// (blockIdx % 8 shares an XCD; MI355X_MICROARCH.md), so with the plain mapping (default)
// the co-resident workgroups of all XCDs work on ~256 CONSECUTIVE elements: the chip
// sweeps every array front to back.  The chunked mapping gives XCD x the contiguous
// element range [x*per_xcd, (x+1)*per_xcd) (one eighth of the pages per L2/TLB).  Which
// one is faster depends on the kernel: the chunked mapping is 0-4 % slower on round 1's
// one-workgroup-per-CU all-streaming shapes (profiles/r01/kbench_xcd_mapping.log) and
// 0.7-1.3 % FASTER on the two-workgroup defaults with the cache window
// (profiles/r02/kbench_np4_nlev72_final.log), so every variant carries its measured
// preference (KernelVariant::prefers_xcd_chunked; the three headline kernels: chunked)
// and caar_set_xcd_chunked overrides it.  Speed only — any placement computes the
// same thing.  Returns -1 for the padding blocks of the rounded-up grid.
__device__ __forceinline__ long long element_of_block(const KernelArgs& k, unsigned b) {
  if (k.per_xcd == 0) return (long long)k.nets + b;
  const unsigned x = b & 7u, s = b >> 3;
  const unsigned e = x * (unsigned)k.per_xcd + s;  // < 8 * per_xcd <= nelem + 7: 32-bit arithmetic is enough
  return (s < (unsigned)k.per_xcd && e < (unsigned)k.nelem) ? (long long)k.nets + e : -1;
}

// First statement of a kernel: one field of every 64-byte line of the 256-byte kernel-argument block is needed HERE (an empty
// asm that takes them as scalar inputs), so the compiler requests all four lines together at the top.  Left alone it requests
// each line where the code first uses it: four scalar-cache misses one after the other in front of a workgroup's first load —
// and that prologue is not hidden (two workgroups per CU): with the two 64-bit divisions that used to sit there as well it cost
// 2 % of the headline (profiles/r05/fixedq_kbench.log, prologue_kbench.log).
__device__ __forceinline__ void request_kernel_args(const KernelArgs& k) {
  asm volatile("" ::"s"(k.D), "s"(k.v), "s"(k.vn0), "s"(k.timelevels));
}

// Hybrid cache policy: is element `ie` (its index in the arrays, 0 .. cache_n-1) one of the cache_count evenly
// spread chosen ones?  (Bresenham: floor((ie+1)*c/n) > floor(ie*c/n).)  A property of the element, not of the
// launch that happens to process it: the budget is the device's, however the host cuts the range into launches.
__device__ __forceinline__ bool element_is_cached(const KernelArgs& k, long long ie) {
  // floor((ie + 1) * c / n) > floor(ie * c / n) with c / n as the 32.32 fixed-point number q the host formed: two scalar
  // multiplies instead of two 64-bit divisions (a v_rcp_f32 sequence of ~100 instructions in front of a workgroup's first
  // load).  The subset differs from the exact Bresenham one by at most one element in count; what matters — an evenly spread
  // subset that is a function of the element's index only — is unchanged.
  // (a hashed, pseudo-random pick of the same density was measured and is no better: docs/EXPERIMENTS.md A)
  const unsigned long long q = ((unsigned long long)k.cache_q_hi << 32) | k.cache_q_lo;
  const unsigned long long e = (unsigned long long)ie;
  return (((e + 1) * q) >> 32) > ((e * q) >> 32);
}

// 1/x for a normal, non-zero fp64 x: v_rcp_f64 seed + two Newton steps (5 instructions,
// <= 1 ulp), instead of the IEEE division sequence (v_div_scale/fmas/fixup, ~12
// instructions and twice the live registers).  The reference divides by p and dp3d
// (P:150,219,291,323); both are positive pressures, so no scaling or fix-up is needed.
// Error budget: DESIGN.md "Numerics".
__device__ __forceinline__ double recip(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}

// a*b + c*d with the contraction spelled out: c*d is rounded, a*b is fused into the sum.  Left to the compiler
// (-ffp-contract=fast) either product may be the fused one, and which one it picks depends on how the operands reached
// the expression (loaded, carried in registers, ...): two instantiations of the same kernel body can then differ in the
// last bit.  With every sum of two products written this way all kernels of one NP — every launch shape, cache policy, the
// step loops with their carried state — perform the same roundings and give bit-identical results.
__device__ __forceinline__ double dot2(double a, double b, double c, double d) { return __builtin_fma(a, b, c * d); }


// Streaming accesses.  Every element array is read once and written once per launch
// and never re-used by another workgroup, so the loads and stores carry the
// non-temporal hint ("nt"): measured +5..8 % bandwidth on the traffic skeleton when
// BOTH loads and stores are non-temporal (profiles/r01/kbench_skeleton_nt.log).
typedef double dbl2 __attribute__((ext_vector_type(2)));
template <bool NT, typename T>
__device__ __forceinline__ T stream_load(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT, typename T>
__device__ __forceinline__ void stream_store(T* p, T v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// Pointer for RE-READING values a kernel parked in LDS to free registers.  The compiler must not replace such a read by the
// register copy of what was stored (that is the point of parking), so the address is laundered through an empty asm; the
// access stays a plain ds_read_b64 in the LDS address space.  (A `volatile` generic pointer does the job too, but compiles to
// flat_load sc0 sc1 + s_waitcnt vmcnt(0): every parked read then also drains all outstanding global loads.)
using lds_cptr = const __attribute__((address_space(3))) double*;
__device__ __forceinline__ lds_cptr lds_reread_ptr(const double* generic_ptr_into_shared) {
  unsigned a = (unsigned)(size_t)(lds_cptr)generic_ptr_into_shared;
  asm volatile("" : "+v"(a));
  return (lds_cptr)(size_t)a;
}

// The kernel arguments are RE-READ from the kernarg segment at the top of every call (scalar loads through a laundered
// pointer): kept alive across the loop, the 16 array pointers and the scalars would hold ~70 SGPRs for the whole kernel,
// the overflow is spilled into VGPR lanes, and the NP=4 two-workgroup shapes no longer fit 256 VGPRs (measured: 270-283
// VGPRs that way; as it stands the kernels are at the register count of their single-call twins).  Used by the step-loop
// kernels (caar_np4_steps.hip, caar_np8.hip), whose first kernel parameter is the KernelArgs.
typedef const __attribute__((address_space(4))) KernelArgs* kernarg_ptr;
__device__ __forceinline__ KernelArgs reload_args() {
  kernarg_ptr kp = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();  // KernelArgs is the first kernel parameter
  asm volatile("" : "+s"(kp));
  KernelArgs k;
#define CAAR_F(f) k.f = kp->f;
  CAAR_F(D) CAAR_F(Dinv) CAAR_F(fcor) CAAR_F(spheremp) CAAR_F(metdet) CAAR_F(rmetdet) CAAR_F(dp3d) CAAR_F(v) CAAR_F(T)
  CAAR_F(phis) CAAR_F(Qdp) CAAR_F(eta_dot_dpdn) CAAR_F(omega_p) CAAR_F(phi) CAAR_F(pecnd) CAAR_F(vn0) CAAR_F(Dvv)
  CAAR_F(hybi) CAAR_F(vadv) CAAR_F(nets) CAAR_F(nelem) CAAR_F(per_xcd) CAAR_F(cache_q_lo) CAAR_F(cache_q_hi) CAAR_F(n0) CAAR_F(np1)
  CAAR_F(nm1) CAAR_F(qn0) CAAR_F(qsize_d) CAAR_F(timelevels) CAAR_F(nlev) CAAR_F(dt2) CAAR_F(rrearth) CAAR_F(eta_ave_w)
  CAAR_F(rv_over_rd_m1) CAAR_F(Rgas) CAAR_F(kappa) CAAR_F(p_top)
#undef CAAR_F
  return k;
}
static_assert(sizeof(KernelArgs) == 18 * 8 + 13 * 4 + 4 /* padding */ + 7 * 8 && sizeof(KernelArgs) == 256, "reload_args lists every member of KernelArgs");

// -DCAAR_DEBUG builds (libcaar_hip_debug.so): the reference's only hot-path assertion, check_dp3d
// (level_vectorized_ppscan/CaarFunctor.hpp:82-97: dp3d(np1) > 0 under !NDEBUG), as a device-side counter instead of an
// abort — a trapping kernel takes the GPU down for everybody on this pool.  Every dp3d(np1) a kernel stores is checked;
// caar_debug_dp3d_violations() (include/caar.h) returns how many were not positive.  (One counter per translation
// unit: the library is built without relocatable device code.)
#ifdef CAAR_DEBUG
static __device__ unsigned long long g_debug_nonpositive_dp3d = 0;
__device__ __forceinline__ void debug_check_dp3d(double dp_np1) {
  if (!(dp_np1 > 0.0)) atomicAdd(&g_debug_nonpositive_dp3d, 1ULL);
}
// host side of this translation unit: read (and optionally clear) the counter
static inline long long debug_dp3d_count_of_this_tu(int reset) {
  unsigned long long n = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_debug_nonpositive_dp3d), sizeof(n)) != hipSuccess) return -1;
  if (reset) {
    const unsigned long long zero = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_debug_nonpositive_dp3d), &zero, sizeof(zero));
  }
  return (long long)n;
}
#else
__device__ __forceinline__ void debug_check_dp3d(double) {}
#endif

// One compiled kernel configuration for a given (np, nlev).
struct KernelVariant {
  const char* kernel;  // demangled kernel name as rocprofv3 prints it
  const char* what;
  hipError_t (*launch)(const KernelArgs&, int num_elems, hipStream_t stream);
  bool prefers_xcd_chunked = false;  // measured faster with each XCD on a contiguous eighth of the element range
  // nsteps calls (time levels rotating between them if `rotate`) as ONE launch, or nullptr: caar_run_steps then replays a
  // graph of single launches
  hipError_t (*launch_steps)(const KernelArgs&, int num_elems, int nsteps, int rotate, hipStream_t stream) = nullptr;
  // the kernel runs the hybrid cache policy (POL = 2): KernelArgs::cache_count elements keep their accumulators in the
  // memory-side cache, 0 makes the same kernel all-streaming — what the adaptive window of caar_abi.hip switches between
  bool hybrid = false;
  // index (same table) of the all-streaming kernel of the same launch shape, or -1: what the adaptive window launches when
  // all-streaming is the policy — the twin holds one code path where the hybrid kernel holds two, and is 1.3 % faster than
  // the hybrid kernel with an empty window (0.3463 against 0.3511 ms, profiles/r04/adaptive_window_demo2.log)
  int streaming_twin = -1;
};

// arguments of the stand-alone operator kernels (caar_operators_ex.hip)
struct OpArgs {
  const double *D, *Dinv, *metdet, *rmetdet, *spheremp, *mp, *metinv, *tensorVisc, *vec_sph2cart;
  const double* dvv;
  const double* in;
  double* out;
  int e0, ne, nlevels;
  double rrearth, alpha, beta, nu_ratio;
};
struct EulerArgs {
  const double *Dinv, *metdet, *rmetdet, *dvv;
  const double *vstar, *qdp;
  double* qtens;
  int e0, ne, nlev, qsize, qsize_d, qn0;
  double dt, rrearth;
};

}  // namespace caar
