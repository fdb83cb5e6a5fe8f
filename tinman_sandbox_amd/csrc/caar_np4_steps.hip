// caar_np4_steps.hip — caar_run_steps / caar_launch_steps as ONE launch for NP=4.
#include <hip/hip_runtime.h>

#include "caar_np4_kernel.h"

namespace caar {

// caar_run_steps as ONE launch (the driver loop main.cpp:113-121 with update_time_levels,
// data_structures.cpp:174-180, between the calls).  Elements are independent and every lane reads and writes only its own
// points of its own element — the state it stores at np1 is what it loads as n0 in the next call, the accumulators are its
// own read-modify-writes — so a workgroup can make all `nsteps` calls for its element back to back with no grid-wide
// barrier: the launch fill and drain (~14 us of a 10 000-element launch, all of a 64-element one) are paid once per
// nsteps calls instead of once per call, and from the second call on the element's arrays are still in the L2 / Infinity
// Cache of the XCD that wrote them (POL = 0: default cache policy; ~95 MB in flight over the chip).  Each call is the same
// code as caar_np4_kernel's (caar_np4_element): bit-identical to nsteps single launches.
// The loop over the calls for one cache policy (SNT / ANT: non-temporal element arrays / accumulators; the step loops come
// with the default policy and all-streaming — the hybrid policy is a single call's: it exists to keep data for the NEXT launch).
// With rotating, distinct time levels (the driver loop) the first call loads everything and every later call takes its n0
// state from registers and (CARRY_LDS) its nm1 state and tracer block from LDS — two instantiations of the body, the
// steady one with no n0 / nm1 / Qdp loads at all.  Without rotation, or with aliased time levels, every call loads
// what it reads (what a lane stored is what it loads: same results, no carry).
template <int NLEV_T, int TPW, int MINW, bool MOIST, bool SNT, bool ANT, int PF, int PARK, int CARRY_LDS, int WAVES>
__device__ __forceinline__ void np4_step_loop(const KernelArgs& k0, int nsteps, int rotate,
                                              Np4Lds<NLEV_T, TPW, false, false, 8, PARK, CARRY_LDS>& lds) {
  int n0 = k0.n0, np1 = k0.np1, nm1 = k0.nm1;
  Np4N0In<TPW> carry;  // dp3d, u, v, T at n0 of this wave's tiles, handed from call to call in registers
  const bool steady = rotate && n0 != np1 && n0 != nm1 && np1 != nm1;  // uniform
  auto rotate_levels = [&] {  // TestData::update_time_levels
    const int t = np1;
    np1 = nm1;
    nm1 = n0;
    n0 = t;
  };
  auto args = [&] {
    KernelArgs k = reload_args();
    k.n0 = n0;
    k.np1 = np1;
    k.nm1 = nm1;
    return k;
  };
  // Stores nothing ever reads: derived_phi is overwritten by every call (only the last one's survives), and with the whole
  // prognostic state travelling on chip (CARRY_LDS) a call's np1 state is read from memory by no later call of the launch
  // — the three time levels must hold what the LAST THREE calls wrote, as after single launches.  The store set is a
  // compile-time property of the steady body (a run-time test around the stores costs 30-60 spilled VGPRs): calls
  // 1 .. nsteps-4 store neither, calls nsteps-3 .. nsteps-2 the state, the last call both.
  if (steady) {
    const int first_mask = ((!CARRY_LDS || nsteps <= 3) ? 1 : 0) | (nsteps == 1 ? 6 : 0);
    caar_np4_element<NLEV_T, TPW, MINW, MOIST, SNT, ANT, PF, false, false, false, 8, PARK, true, CARRY_LDS, 0, -1, WAVES>(args(), lds, &carry, first_mask);
    constexpr int CIN = CARRY_LDS == 4 ? 11 : (CARRY_LDS >= 2 ? 15 : (CARRY_LDS ? 7 : 1));
    // Between two calls: nothing a wave reads in LDS was written by another wave except the tile totals, and with state
    // carried in LDS (CARRY_LDS) those exist twice and alternate calls use alternate sets — a wave that is done with call s
    // starts call s+1 without waiting for the others (they meet at the barrier behind its first phase; by the time anyone
    // writes a set again, two barriers later, everybody has left the phase that read it).  Without the LDS carry the sets
    // do not fit next to the parked scan results: one barrier.
    auto between = [&] {
      rotate_levels();
      if constexpr (!CARRY_LDS) wg_barrier<true>();
    };
    int s = 1;
    if constexpr (CARRY_LDS) {
      for (; s < nsteps - 3; ++s) {
        between();
        caar_np4_element<NLEV_T, TPW, MINW, MOIST, SNT, ANT, PF, false, false, false, 8, PARK, true, CARRY_LDS, CIN, 0, WAVES>(args(), lds, &carry, 3, s & 1);
      }
    }
    for (; s < nsteps - 1; ++s) {
      between();
      caar_np4_element<NLEV_T, TPW, MINW, MOIST, SNT, ANT, PF, false, false, false, 8, PARK, true, CARRY_LDS, CIN, 1, WAVES>(args(), lds, &carry, 3, s & 1);
    }
    if (s < nsteps) {
      between();
      caar_np4_element<NLEV_T, TPW, MINW, MOIST, SNT, ANT, PF, false, false, false, 8, PARK, true, CARRY_LDS, CIN, 7, WAVES>(args(), lds, &carry, 3, s & 1);
    }
  } else {
    for (int s = 0; s < nsteps; ++s) {  // (no carry: every call stores everything except a phi that will be overwritten)
      caar_np4_element<NLEV_T, TPW, MINW, MOIST, SNT, ANT, PF, false, false, false, 8, PARK, true, CARRY_LDS, 0, -1, WAVES>(args(), lds, &carry, 1 | 4 | (s == nsteps - 1 ? 2 : 0));
      if (rotate) rotate_levels();
      wg_barrier<true>();
    }
  }
}

// CARRY_LDS: the nm1 state and the tracer block travel from call to call in LDS (NLEV=72: 46 KB more, two workgroups per CU
// still fit; NLEV=128 has no room next to its parked scan results).
template <int NLEV_T, int TPW, int MINW, bool MOIST, int POL, int PF, int PARK, int CARRY_LDS, int WAVES>
__global__ __launch_bounds__((WAVES ? WAVES : ((NLEV_T + 3) / 4 + TPW - 1) / TPW) * 64, MINW) void caar_np4_steps_kernel(const KernelArgs k0, int nsteps, int rotate) {
  __shared__ Np4Lds<NLEV_T, TPW, false, false, 8, PARK, CARRY_LDS> lds;
  const long long ie_s = element_of_block(k0, blockIdx.x);
  if (ie_s < 0) return;
  static_assert(POL == 0 || POL == 1, "step loops: default cache policy or all-streaming (the hybrid policy is a single call's)");
  np4_step_loop<NLEV_T, TPW, MINW, MOIST, POL == 1, POL == 1, PF, PARK, CARRY_LDS, WAVES>(k0, nsteps, rotate, lds);
}

template <int NLEV, int TPW, int MINW, int POL, int PF, int PARK, int CARRY_LDS = (NLEV <= 80 ? 1 : 0), int WAVES = 0>
static hipError_t launch_np4_steps(const KernelArgs& k, int num_elems, int nsteps, int rotate, hipStream_t stream) {
  constexpr int THREADS = (WAVES ? WAVES : ((NLEV + 3) / 4 + TPW - 1) / TPW) * 64;
  if (k.vadv) return hipErrorNotSupported;  // the Eulerian form steps through the graph of single launches
  const int grid = k.per_xcd ? 8 * k.per_xcd : num_elems;
  if (k.qn0 >= 0)
    hipLaunchKernelGGL((caar_np4_steps_kernel<NLEV, TPW, MINW, true, POL, PF, PARK, CARRY_LDS, WAVES>), dim3(grid), dim3(THREADS), 0, stream, k, nsteps, rotate);
  else
    hipLaunchKernelGGL((caar_np4_steps_kernel<NLEV, TPW, MINW, false, POL, PF, PARK, CARRY_LDS, WAVES>), dim3(grid), dim3(THREADS), 0, stream, k, nsteps, rotate);
  return hipGetLastError();
}

// the instantiations the variant tables of caar_np4.hip point to
// What the NLEV=72 two-workgroup loop carries in LDS from call to call.  4 (default since round 4): the nm1 state AND the
// accumulators vn0, omega_p — 7 slots, 80.2 KB per workgroup, the most that still lets two workgroups share a CU; a steady
// call then reads only the tracer block and pecnd (18 KB per element instead of 64 KB read + written) and writes nothing.
// (Rounds 3-4a carried 1: the nm1 state and the tracer block, the accumulators through the L2 / Infinity Cache.)  Same speed with
// the default cache policy (0.1046 against 0.1050 ms per call at 10 000 elements: the loop is bound by instruction issue and
// latency at two waves per SIMD, not by memory) — but no longer dependent on it (the all-streaming loop: 0.1218 -> 0.1053) or
// on what else uses the caches; bit-identical (profiles/r04/carry4_steps.log, carry4_bits.log).
constexpr int kSteps72Carry = 4;
#define CAAR_STEPS(NLEV, TPW, MINW, POL, PF, PARK)                                                                  \
  hipError_t launch_np4_steps_##NLEV##_##POL(const KernelArgs& k, int num_elems, int nsteps, int rotate, hipStream_t s) { \
    return launch_np4_steps<NLEV, TPW, MINW, POL, PF, PARK, (NLEV == 72 ? kSteps72Carry : (NLEV <= 80 ? 1 : 0))>(k, num_elems, nsteps, rotate, s); \
  }
// NLEV=72: FOUR waves with 5, 5, 4, 4 of the 18 tiles (caar_np4_kernel.h UNEVEN), two workgroups per CU: the loop is bound by
// instruction issue, and two 3 x 6 workgroups put 2, 2, 1, 1 waves on a CU's four SIMDs (12 tiles per pair of elements on
// the busiest) where two 4-wave ones put one wave each on every SIMD (9-10): 0.134 against 0.151 ms per call at 10 000
// elements (profiles/r03/steps_bench_72_4w.log, simd_placement_probe.log).  Nothing parked: five tiles fit the registers.
CAAR_STEPS(72, 5, 2, 1, 0, 0)
CAAR_STEPS(72, 5, 2, 0, 0, 0)
CAAR_STEPS(128, 8, 2, 1, 0, 27)
CAAR_STEPS(128, 8, 2, 0, 0, 27)
// Other level counts whose carried state fits two workgroups per CU (5 slots x tiles x 512 B + the tile totals <= 80 KB):
// NLEV=80 (E3SM's 80-level configuration: 4 waves x 5 tiles, 68 KB), 64 (4 x 4), 60 (4, 4, 4, 3).  NLEV=96 (82 KB) and the
// level counts that are not a multiple of 4 keep the hipGraph of single launches.
#if CAAR_EXTRA_NLEV
CAAR_STEPS(80, 5, 2, 0, 0, 0)
CAAR_STEPS(64, 4, 2, 0, 0, 0)
CAAR_STEPS(60, 4, 2, 0, 0, 0)
#endif
#undef CAAR_STEPS
// What the default variants use.  From the second call on a call's inputs are the previous call's outputs: n0 state in
// registers, (NLEV=72) nm1 state and tracer block in LDS, the rest still on chip if the accesses use the DEFAULT cache
// policy — the opposite of what a single call wants.  NLEV=72: default policy at every size (0.134 ms per call at 10 000
// elements against 0.322 for single launches).  profiles/r03/steps_bench_72_4w_thresh.log, steps_bench_128_m2.log.
hipError_t launch_np4_steps_72_onchip(const KernelArgs& k, int num_elems, int nsteps, int rotate, hipStream_t s);
hipError_t launch_np4_steps_72_auto(const KernelArgs& k, int num_elems, int nsteps, int rotate, hipStream_t s) {
  // Up to one element per CU a call is bound by the latency of ONE element's step: everything on chip (one 8-wave workgroup
  // per CU; 4.7-4.9 us per call at 64-256 elements against 5.5-5.7).  Beyond that two 4-wave workgroups per CU, overlapping two
  // elements, win (7.6 against 9.9 us at 384 elements, 0.134 against 0.17-0.19 ms at 10 000).
  // profiles/r03/steps_bench_72_4w_thresh.log.
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  return num_elems <= cus ? launch_np4_steps_72_onchip(k, num_elems, nsteps, rotate, s)
                          : launch_np4_steps_72_0(k, num_elems, nsteps, rotate, s);
}
// NLEV=72 with EVERYTHING a call needs from its predecessor on chip (CARRY_LDS = 2: the accumulators and pecnd in LDS as
// well, the metric terms left staged): one workgroup per CU, 101 KB of LDS.  A steady call then touches no element array
// in memory.  EIGHT waves with 3, 3, 2, 2, 2, 2, 2, 2 of the 18 tiles: two waves on every SIMD with 5, 5, 4, 4 tiles between
// them, where 6 waves x 3 tiles put 6, 3, 3, 6 (4.7-4.9 against 5.1-5.2 us per call at 64-256 elements,
// profiles/r03/steps_bench_72_onchip8.log).
hipError_t launch_np4_steps_72_onchip(const KernelArgs& k, int num_elems, int nsteps, int rotate, hipStream_t s) {
  return launch_np4_steps<72, 3, 2, 0, 0, 0, 2, 8>(k, num_elems, nsteps, rotate, s);
}

// NLEV=128: ONE workgroup per CU, 8 waves x 4 tiles, with the room that leaves in LDS used to carry the nm1 state and the
// tracer block like NLEV=72 does (96 KB: 0.356 ms per call at 12 500 elements against 0.515 for the two-workgroup 4 x 8
// shape, whose LDS is full of parked scan results and which therefore reads nm1 from cache;
// profiles/r03/steps_bench_128_8x4.log) and vn0 and omega_p as well (CARRY_LDS = 3, 145 KB: 0.339 ms per call,
// profiles/r03/steps_bench_128_m2.log).  Default cache policy.
hipError_t launch_np4_steps_128_auto(const KernelArgs& k, int num_elems, int nsteps, int rotate, hipStream_t s) {
  // CARRY_LDS = 3: vn0 and omega_p are carried in LDS too (145 KB; eta_dot_dpdn and pecnd still come from cache)
  return launch_np4_steps<128, 4, 2, 0, 0, 0, 3>(k, num_elems, nsteps, rotate, s);
}


#ifdef CAAR_DEBUG
long long debug_dp3d_count_np4_steps(int reset) { return debug_dp3d_count_of_this_tu(reset); }
#endif

}  // namespace caar
