// caar_np4.hip — compute_and_apply_rhs for NP=4 on gfx950 (MI355X), hand-written HIP.
//
// Replaces, for one element per workgroup, the whole body of the reference's
// element loop (compute_and_apply_rhs_test/cxx/pointers_only/compute_and_apply_rhs.cpp:74-258,
// Fortran fortran/routine_mod.F90:69-191) including the three sphere operators
// (sphere_operators.cpp:9-129) and the two vertical integrals (P:280-352).
//
// Mapping:
//   * one workgroup = one element; one wavefront = TPW "tiles"; one tile = 4
//     consecutive levels x 16 GLL points = 64 lanes (lane = 16a + 4lev + b, the operand layout of v_mfma_f64_4x4x4).
//     In the reference layout [lev][a][b] that is 64 consecutive doubles, so every
//     field access of a wave covers one contiguous 512 B (scalar) / 1 KiB (v)
//     segment and each input byte is read exactly once, each output written once.
//   * the NP x NP Dvv contractions of a tile's four levels are ONE v_mfma_f64_4x4x4 each (caar_np4_ops.h "MFMA form").
//     No cross-wave dependency.
//   * Dvv and the element's metric terms (D, Dinv, metdet, rmetdet, fcor, spheremp,
//     phis) are staged once per workgroup in LDS.
//   * the three vertical integrals (pressure, geopotential, omega) are blocked
//     scans: in-wave over the 4 levels of a tile, tile totals through LDS, two
//     workgroup barriers.  Everything else stays in registers from load to store.
//   * every array is streamed exactly once per launch (non-temporal loads/stores; measured HBM-side traffic =
//     algorithmic bytes x1.0002), except that the default kernels keep the three read-modify-write accumulators of
//     part of the elements in the 256 MB Infinity Cache between calls (POL = 2).
//   * the default launch shapes put TWO workgroups of FOUR waves (two elements in different phases, one wave of each on
//     every SIMD of the CU) on a CU: NLEV=72 4 waves owning 5, 5, 4, 4 of the 18 tiles (240 VGPRs; three-wave workgroups
//     load the four SIMDs 2, 2, 1, 1), NLEV=128 4 waves x 8 tiles with four scan results parked in LDS (PARK).  With the
//     cache window that is worth 76 -> 84-85 % of the HBM peak at NLEV=72.
#include <hip/hip_runtime.h>

#include "caar_np4_kernel.h"

namespace caar {

// caar_np4_steps.hip: nsteps calls as one launch, per (NLEV, cache policy)
#define CAAR_STEPS_DECL(NLEV, POL) hipError_t launch_np4_steps_##NLEV##_##POL(const KernelArgs&, int, int, int, hipStream_t)
CAAR_STEPS_DECL(72, 1);
CAAR_STEPS_DECL(72, 0);
CAAR_STEPS_DECL(128, 1);
CAAR_STEPS_DECL(128, 0);
#if CAAR_EXTRA_NLEV
CAAR_STEPS_DECL(80, 0);
CAAR_STEPS_DECL(64, 0);
CAAR_STEPS_DECL(60, 0);
#endif
#undef CAAR_STEPS_DECL
hipError_t launch_np4_steps_72_auto(const KernelArgs&, int, int, int, hipStream_t);   // cache policy by footprint
hipError_t launch_np4_steps_128_auto(const KernelArgs&, int, int, int, hipStream_t);

// explicit instantiations + launchers --------------------------------------------------
static int cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// VTPW/VMINW/VPF/VPARK: the launch shape of the Eulerian (rsplit == 0) form, which holds more live values per
// level: the two-workgroup shapes need part of them parked in LDS (VPARK; bit 32 re-reads u, v, T from the column
// copy the vertical advection keeps there anyway).  tools/eulerian_bench.py, tools/probes/eulerian_variants.py.
// NLEV=72: FOUR waves with 5, 5, 4, 4 tiles (VTPW = 5; caar_np4_kernel.h UNEVEN): the Eulerian form is heavy on instruction
// issue, and two 3-wave workgroups load a CU's four SIMDs 2, 2, 1, 1 where two 4-wave ones load them evenly: 76.8 -> 79.9-80.3 %
// (profiles/r03/eulerian_bench_4w.log; five tiles also fit 229 registers instead of 253).
// With the contractions on the matrix cores (~14 registers fewer) five tiles need no parking at all (VPARK = 0: 247 VGPRs,
// 82.0 against 80.4 % with p / divdp prefix / divdp parked), and NLEV=128 fits its two-workgroup shape at last: 4 waves x 8
// tiles with ONE scan result (p) parked and u, v, T re-read from the column copy (VPARK = 33: 81.4 KB of LDS, 22 VGPRs
// spilled) — 77.8 against 72.9 % for 8 waves x 4 tiles, one workgroup per CU (profiles/r03/eulerian_vpark.log).
// VPOL: the cache policy of the Eulerian form (the NLEV=128 comparator variants all take the hybrid one: one spilling
// instantiation in the library instead of three).
template <int NLEV, int TPW, int MINW, int POL, int PF = 0, int PERSIST_WG_PER_CU = 0, bool ETA_COND = false,
          int VTPW = TPW, int VMINW = MINW, int VPF = PF, int PARK = 0, int VPARK = 0, int VPOL = POL>
static hipError_t launch_np4(const KernelArgs& k, int num_elems, hipStream_t stream) {
  constexpr int THREADS = ((NLEV + 3) / 4 + TPW - 1) / TPW * 64;
  constexpr bool PERSIST = PERSIST_WG_PER_CU > 0;
  int grid = k.per_xcd ? 8 * k.per_xcd : num_elems;
  if (PERSIST) {
    grid = cu_count() * PERSIST_WG_PER_CU;
    if (grid > num_elems) grid = num_elems;
  }
  if (k.vadv) {  // rsplit == 0: the plain (non-persistent, unconditional eta store) form
    constexpr int VTHREADS = ((NLEV + 3) / 4 + VTPW - 1) / VTPW * 64;
    if (PERSIST) grid = k.per_xcd ? 8 * k.per_xcd : num_elems;
    if (k.qn0 >= 0)
      hipLaunchKernelGGL((caar_np4_kernel<NLEV, VTPW, VMINW, true, VPOL, VPF, false, false, true, 8, VPARK>), dim3(grid), dim3(VTHREADS), 0, stream, k);
    else
      hipLaunchKernelGGL((caar_np4_kernel<NLEV, VTPW, VMINW, false, VPOL, VPF, false, false, true, 8, VPARK>), dim3(grid), dim3(VTHREADS), 0, stream, k);
    return hipGetLastError();
  }
  if (k.qn0 >= 0)
    hipLaunchKernelGGL((caar_np4_kernel<NLEV, TPW, MINW, true, POL, PF, PERSIST, ETA_COND, false, 8, PARK>), dim3(grid), dim3(THREADS), 0, stream, k);
  else  // dry branch (P:128-139): the Qdp block is never touched
    hipLaunchKernelGGL((caar_np4_kernel<NLEV, TPW, MINW, false, POL, PF, PERSIST, ETA_COND, false, 8, PARK>), dim3(grid), dim3(THREADS), 0, stream, k);
  return hipGetLastError();
}

// Tuning variants (caar_select_variant): TPW = tiles per wave, MINW = waves per SIMD the
// register allocator must leave room for (=> workgroups per CU), NT = non-temporal
// streaming accesses.  Index 0 is the default.
// (non-const on purpose: const globals are also emitted for the device, where the host
// launchers they point to do not exist)
KernelVariant kNp4Nlev72[] = {
    {"caar_np4_kernel<72, 5, 1, true, 2, 0, false, false, false, 8, 0>", "4 waves x 5, 5, 4, 4 tiles (two workgroups per CU, one wave of each on every SIMD), hybrid cache policy (nt; the accumulators of every n-th element stay in the Infinity Cache), update loads one tile ahead", launch_np4<72, 5, 1, 2, 0, 0, false, 5, 2, 0, 0, 0>, true, launch_np4_steps_72_auto, true, 1},
    {"caar_np4_kernel<72, 5, 1, true, 1, 0, false, false, false, 8, 0>", "4 waves x 5, 5, 4, 4 tiles (two workgroups per CU), nt (all streaming), update loads one tile ahead", launch_np4<72, 5, 1, true, 0, 0, false, 5, 2, 0, 0, 0>, false, launch_np4_steps_72_1},
    {"caar_np4_kernel<72, 6, 1, true, 2, 0, false, false, false, 8, 0>", "3 waves x 6 tiles (two workgroups per CU: 2, 2, 1, 1 waves on the four SIMDs; the default of rounds 2-3), hybrid cache policy, update loads one tile ahead", launch_np4<72, 6, 1, 2, 0, 0, false, 6, 2, 0, 0, 43>, true, nullptr, true},
    {"caar_np4_kernel<72, 2, 1, true, 1, 1, false, false, false, 8, 0>", "9 waves x 2 tiles, nt, update loads before the last barrier", launch_np4<72, 2, 1, true, 1, 0, false, 3, 2, 0>},
    {"caar_np4_kernel<72, 5, 1, true, 2, 0, false, true, false, 8, 0>", "the default's shape and cache policy; eta_dot_dpdn stored only where its bits change (skips the no-op write-back of the vertically Lagrangian form: NOT the contract traffic, same results)", launch_np4<72, 5, 1, 2, 0, 0, true, 5, 2, 0, 0, 0>, true, nullptr, true},
    {"caar_np4_kernel<72, 3, 2, true, 1, 1, true, false, false, 8, 0>", "persistent (1 workgroup/CU), 6 waves x 3 tiles, nt", launch_np4<72, 3, 2, true, 1, 1>},
    {"caar_np4_kernel<72, 5, 1, true, 0, 0, false, false, false, 8, 0>", "4 waves x 5, 5, 4, 4 tiles (two workgroups per CU), default cache policy (nothing streams: what a fused multi-step launch wants)", launch_np4<72, 5, 1, 0, 0, 0, false, 5, 2, 0, 0, 0>, true, launch_np4_steps_72_0},
};
int kNp4Nlev72Count = sizeof(kNp4Nlev72) / sizeof(kNp4Nlev72[0]);

KernelVariant kNp4Nlev128[] = {
    {"caar_np4_kernel<128, 8, 2, true, 2, 0, false, false, false, 8, 27>", "4 waves x 8 tiles (two workgroups per CU), p / divdp prefix / divdp / T_v parked in LDS between the phases, hybrid cache policy (nt; the accumulators of every n-th element stay in the Infinity Cache), update loads one tile ahead", launch_np4<128, 8, 2, 2, 0, 0, false, 8, 2, 0, 27, 33>, true, launch_np4_steps_128_auto, true, 1},
    {"caar_np4_kernel<128, 8, 2, true, 1, 0, false, false, false, 8, 27>", "4 waves x 8 tiles (two workgroups per CU), p / divdp prefix / divdp / T_v parked in LDS, nt (all streaming), update loads one tile ahead", launch_np4<128, 8, 2, true, 0, 0, false, 8, 2, 0, 27, 33, 2>, false, launch_np4_steps_128_1},
    {"caar_np4_kernel<128, 4, 2, true, 2, 1, false, false, false, 8, 0>", "8 waves x 4 tiles, hybrid cache policy (nt; the accumulators of every n-th element stay in the Infinity Cache), update loads before the last barrier", launch_np4<128, 4, 2, 2, 1, 0, false, 8, 2, 0, 0, 33>, false, nullptr, true},
    {"caar_np4_kernel<128, 4, 2, true, 1, 1, false, false, false, 8, 0>", "8 waves x 4 tiles, nt, update loads before the last barrier", launch_np4<128, 4, 2, true, 1, 0, false, 8, 2, 0, 0, 33, 2>},
    {"caar_np4_kernel<128, 8, 2, true, 0, 0, false, false, false, 8, 27>", "4 waves x 8 tiles (two workgroups per CU), scan results parked in LDS, default cache policy (what a fused multi-step launch wants)", launch_np4<128, 8, 2, 0, 0, 0, false, 8, 2, 0, 27, 33, 2>, true, launch_np4_steps_128_0},
};
int kNp4Nlev128Count = sizeof(kNp4Nlev128) / sizeof(kNp4Nlev128[0]);

// Other level counts HOMME configurations use (the reference builds any PLEV from config.h):
// one launch shape each, same kernel template.  -DCAAR_EXTRA_NLEV=1 builds only (caar_kernel_args.h): by default these level
// counts run through the run-time-level-count kernel below.
#if CAAR_EXTRA_NLEV
KernelVariant kNp4Nlev32[] = {
    {"caar_np4_kernel<32, 2, 1, true, 2, 1, false, false, false, 8, 0>", "4 waves x 2 tiles, room for 1 wave/SIMD, hybrid cache policy", launch_np4<32, 2, 1, 2, 1>},
    {"caar_np4_kernel<32, 2, 1, true, 1, 1, false, false, false, 8, 0>", "4 waves x 2 tiles, room for 1 wave/SIMD, nt", launch_np4<32, 2, 1, true, 1>},
};
int kNp4Nlev32Count = sizeof(kNp4Nlev32) / sizeof(kNp4Nlev32[0]);
KernelVariant kNp4Nlev60[] = {
    {"caar_np4_kernel<60, 4, 1, true, 2, 0, false, false, false, 8, 0>", "4 waves x 4, 4, 4, 3 tiles (two workgroups per CU), hybrid cache policy", launch_np4<60, 4, 1, 2, 0, 0, false, 4, 2, 0, 0, 0>, true, launch_np4_steps_60_0},
    {"caar_np4_kernel<60, 4, 1, true, 1, 0, false, false, false, 8, 0>", "4 waves x 4, 4, 4, 3 tiles (two workgroups per CU), nt", launch_np4<60, 4, 1, true, 0, 0, false, 4, 2, 0, 0, 0>},
};
int kNp4Nlev60Count = sizeof(kNp4Nlev60) / sizeof(kNp4Nlev60[0]);
KernelVariant kNp4Nlev64[] = {
    {"caar_np4_kernel<64, 4, 2, true, 2, 1, false, false, false, 8, 0>", "4 waves x 4 tiles, room for 2 waves/SIMD, hybrid cache policy", launch_np4<64, 4, 2, 2, 1>, false, launch_np4_steps_64_0},
    {"caar_np4_kernel<64, 4, 2, true, 1, 1, false, false, false, 8, 0>", "4 waves x 4 tiles, room for 2 waves/SIMD, nt", launch_np4<64, 4, 2, true, 1>},
};
int kNp4Nlev64Count = sizeof(kNp4Nlev64) / sizeof(kNp4Nlev64[0]);
KernelVariant kNp4Nlev80[] = {
    {"caar_np4_kernel<80, 5, 1, true, 2, 0, false, false, false, 8, 0>", "4 waves x 5 tiles (two workgroups per CU), hybrid cache policy", launch_np4<80, 5, 1, 2, 0, 0, false, 5, 2, 0, 0, 0>, true, launch_np4_steps_80_0},
    {"caar_np4_kernel<80, 5, 1, true, 1, 0, false, false, false, 8, 0>", "4 waves x 5 tiles (two workgroups per CU), nt", launch_np4<80, 5, 1, true, 0, 0, false, 5, 2, 0, 0, 0>},
};
int kNp4Nlev80Count = sizeof(kNp4Nlev80) / sizeof(kNp4Nlev80[0]);
KernelVariant kNp4Nlev96[] = {
    {"caar_np4_kernel<96, 6, 1, true, 2, 0, false, false, false, 8, 0>", "4 waves x 6 tiles (two workgroups per CU), hybrid cache policy", launch_np4<96, 6, 1, 2, 0, 0, false, 6, 2, 0, 0, 32>, true},
    {"caar_np4_kernel<96, 6, 1, true, 1, 0, false, false, false, 8, 0>", "4 waves x 6 tiles (two workgroups per CU), nt", launch_np4<96, 6, 1, true, 0, 0, false, 6, 2, 0, 0, 32>},
};
int kNp4Nlev96Count = sizeof(kNp4Nlev96) / sizeof(kNp4Nlev96[0]);
#endif
// Any other level count up to 256: the kernel with a run-time level count (NLEV_T = 0).
// EUL: whether the Eulerian (rsplit == 0) form is compiled for this shape (it is not where it spills registers)
template <int TPW, int MAXW, int PF, int VPF = 0, int PARK = 0, bool EUL = true>
static hipError_t launch_np4_dyn_shape(const KernelArgs& k, int num_elems, hipStream_t stream) {
  const int tiles = (k.nlev + 3) / 4, waves = (tiles + TPW - 1) / TPW;
  const int grid = k.per_xcd ? 8 * k.per_xcd : num_elems;
  const dim3 block(waves * 64);
  if (waves > MAXW) return hipErrorInvalidValue;
  if (k.vadv) {
    if constexpr (EUL) {
      if (k.qn0 >= 0) hipLaunchKernelGGL((caar_np4_kernel<0, TPW, 1, true, true, VPF, false, false, true, MAXW>), dim3(grid), block, 0, stream, k);
      else hipLaunchKernelGGL((caar_np4_kernel<0, TPW, 1, false, true, VPF, false, false, true, MAXW>), dim3(grid), block, 0, stream, k);
    } else {
      return hipErrorNotSupported;
    }
  } else {
    if (k.qn0 >= 0) hipLaunchKernelGGL((caar_np4_kernel<0, TPW, 1, true, 2, PF, false, false, false, MAXW, PARK>), dim3(grid), block, 0, stream, k);
    else hipLaunchKernelGGL((caar_np4_kernel<0, TPW, 1, false, 2, PF, false, false, false, MAXW, PARK>), dim3(grid), block, 0, stream, k);
  }
  return hipGetLastError();
}
static hipError_t launch_np4_dyn(const KernelArgs& k, int num_elems, hipStream_t stream) {
  if (k.nlev < 2 || k.nlev > 256) return hipErrorInvalidValue;
  // FOUR waves where the tiles per wave stay within the registers of this masked form (<= 6 with the MFMA contractions:
  // 248 VGPRs; the Eulerian form: <= 5, it spills at 6): workgroups of four waves land one wave on every SIMD of a CU, other
  // counts load the SIMDs unevenly (tools/probes/simd_placement_probe.hip)
  const int tiles = (k.nlev + 3) / 4;
  if (tiles <= 8) return launch_np4_dyn_shape<2, 8, 1>(k, num_elems, stream);    // <= 4 waves x 2 tiles
  if (tiles <= 12) return launch_np4_dyn_shape<3, 8, 0>(k, num_elems, stream);   // 4 waves x 3
  if (tiles <= 16) return launch_np4_dyn_shape<4, 8, 0>(k, num_elems, stream);   // 4 waves x 4
  if (tiles <= 20) return launch_np4_dyn_shape<5, 8, 0>(k, num_elems, stream);   // 4 waves x 5
  if (tiles <= 32 && k.vadv) return launch_np4_dyn_shape<4, 8, 0>(k, num_elems, stream);   // Eulerian form: <= 8 waves x 4 tiles (no spills)
  if (tiles <= 24) return launch_np4_dyn_shape<6, 8, 0, 0, 0, false>(k, num_elems, stream);   // 4 waves x 6
  if (tiles <= 32) return launch_np4_dyn_shape<8, 4, 0, 0, 27, false>(k, num_elems, stream);  // 4 waves x 8 tiles, scan results parked: two workgroups per CU like the NLEV=128 kernel
  // beyond 128 levels: <= 8 waves x 8 tiles, one workgroup per CU, p / divdp prefix / divdp / T_v parked in LDS like the
  // NLEV=128 kernel (157 KB; 231 VGPRs).  Its Eulerian form spills 65-90 VGPRs: -DCAAR_EXTRA_NLEV=1 builds only, else the
  // call fails with hipErrorNotSupported (rsplit == 0 is served up to 128 levels)
  return launch_np4_dyn_shape<8, 8, 0, 0, 27, CAAR_EXTRA_NLEV != 0>(k, num_elems, stream);
}
KernelVariant kNp4NlevAny[] = {{"caar_np4_kernel<0, ...>", "run-time level count (2..256): four waves x 2..6 tiles up to 96 levels, four waves x 8 tiles (scan results parked) up to 128, 8 waves x 8 (parked) beyond, dead rows masked, hybrid cache policy", launch_np4_dyn, false, nullptr, true}};

// level counts that are not a multiple of 4 (last tile partly empty)
#if CAAR_EXTRA_NLEV
KernelVariant kNp4Nlev26[] = {
    {"caar_np4_kernel<26, 2, 1, true, 2, 1, false, false, false, 8, 0>", "4 waves x 2, 2, 2, 1 tiles (the last one half empty), hybrid cache policy", launch_np4<26, 2, 1, 2, 1>, true},
    {"caar_np4_kernel<26, 2, 1, true, 1, 1, false, false, false, 8, 0>", "4 waves x 2, 2, 2, 1 tiles, nt", launch_np4<26, 2, 1, true, 1>},
};
int kNp4Nlev26Count = sizeof(kNp4Nlev26) / sizeof(kNp4Nlev26[0]);
KernelVariant kNp4Nlev30[] = {
    {"caar_np4_kernel<30, 2, 2, true, 2, 1, false, false, false, 8, 0>", "4 waves x 2 tiles, room for 2 waves/SIMD, hybrid cache policy", launch_np4<30, 2, 2, 2, 1>},
    {"caar_np4_kernel<30, 2, 2, true, 1, 1, false, false, false, 8, 0>", "4 waves x 2 tiles, room for 2 waves/SIMD, nt", launch_np4<30, 2, 2, true, 1>},
};
int kNp4Nlev30Count = sizeof(kNp4Nlev30) / sizeof(kNp4Nlev30[0]);
#endif

#ifdef CAAR_DEBUG
long long debug_dp3d_count_np4(int reset) { return debug_dp3d_count_of_this_tu(reset); }
#endif

}  // namespace caar
