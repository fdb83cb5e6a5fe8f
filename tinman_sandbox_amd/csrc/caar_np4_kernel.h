// caar_np4_kernel.h — the NP=4 kernel templates (element body, single-call kernel) shared by caar_np4.hip (launch
// shapes and variant tables) and caar_np4_steps.hip (the step-loop kernel of caar_run_steps).  See caar_np4.hip for the
// mapping.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "caar_kernel_args.h"
#include "caar_np4_ops.h"

namespace caar {

// ------------------------------------------------- in-wave scans over the 4 sub-levels
// Lanes l, l+16, l+32, l+48 hold levels 4t..4t+3 of one GLL point.
__device__ __forceinline__ double shfl_abs(double x, int src_lane) { return __shfl(x, src_lane, 64); }

// Lane -> (level inside the tile, GLL point): lane = 16a + 4lev + b, the operand / result layout of v_mfma_f64_4x4x4
// (caar_np4_ops.h "MFMA form"): the contractions of the four levels of a tile are ONE matrix instruction each.  A wave covers
// 64 consecutive doubles of a field block; LSTEP is the lane distance between consecutive levels (the in-tile scans).  The
// whole NP=4 family uses this ONE form, so that every kernel of it — launch shapes, cache policies, the step loops — stays
// bit-identical to every other.  (Rounds 1-3 had lane = 16lev + 4a + b with the contractions inside DPP rows; the A/B that
// retired it: docs/EXPERIMENTS.md A, CAAR_NP4_MFMA.)
constexpr int LSTEP = 4;
using Np4Ctx = Mfma4Ctx;

// The scan partners lane -/+ 4, -/+ 8 sit in the lane's own 16-lane row, so the moves are DPP row shifts (row_shr / row_shl:
// v_mov_b32_dpp, two per fp64 value, no LDS traffic, no lgkmcnt wait) instead of ds_bpermute pairs.  Pure data movement:
// results are bit-identical to the __shfl form.  A lane without a partner (the row shift runs off the row) must add
// NOTHING: its addend is -0.0 — x + (-0.0) == x for every x, signed zeros included, which x + 0.0 is not — built from the
// halves: the low word with bound_ctrl (0 where there is no source lane), the high word keeps `old` = 0x80000000.
// Which kernels take the DPP form — measured, two builds alternating on one box (profiles/r04/dppscan_kbench.log,
// dppscan_steps.log; 405 bit fingerprints of tools/ab_bits.py agree between the builds): NLEV=72 single call 83.6-84.3 ->
// 86.0-87.3 % of peak (all-streaming 75-75.7 -> 76.4-78.2), step loops 0.1236 -> 0.1049 ms per call at NLEV=72 and
// 0.2969 -> 0.2535 at NLEV=128 (-15 %: the scans' three dependent ds_bpermute round trips per integral were the longest
// latency chain of a tile).  The single call at NLEV=128 did not gain (81.2 -> 80.9 %, its all-streaming twin 76.8 -> 75.6):
// it keeps the __shfl form.
template <int NLEV_T, bool STEPS>
constexpr bool np4_scan_dpp() { return !(NLEV_T == 128 && !STEPS); }
enum { DPP_ROW_SHL = 0x100, DPP_ROW_SHR = 0x110 };
// x of the lane CTRL names, or (lanes without a source) -0.0 if NEG_ZERO else +0.0
template <int CTRL, bool NEG_ZERO>
__device__ __forceinline__ double dpp_row_shift(double x) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, true);
  const int hi = NEG_ZERO ? __builtin_amdgcn_update_dpp((int)0x80000000u, (int)(b >> 32), CTRL, 0xf, 0xf, false)
                          : __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// inclusive prefix (towards higher levels) and the matching exclusive value
template <bool DPP>
__device__ __forceinline__ void scan_down(double x, int lane, int sub, double& incl, double& excl) {
  if constexpr (DPP) {
    x += dpp_row_shift<DPP_ROW_SHR + LSTEP, true>(x);      // sub >= 1: + level above
    x += dpp_row_shift<DPP_ROW_SHR + 2 * LSTEP, true>(x);  // sub >= 2: + the pair above
    incl = x;
    excl = dpp_row_shift<DPP_ROW_SHR + LSTEP, false>(x);   // sub == 0: 0.0
    return;
  }
  double t = shfl_abs(x, lane - LSTEP);
  if (sub >= 1) x += t;
  t = shfl_abs(x, lane - 2 * LSTEP);
  if (sub >= 2) x += t;
  incl = x;
  t = shfl_abs(x, lane - LSTEP);
  excl = sub >= 1 ? t : 0.0;
}
// inclusive suffix (towards lower level index) and the matching exclusive value
template <bool DPP>
__device__ __forceinline__ void scan_up(double x, int lane, int sub, double& incl, double& excl) {
  if constexpr (DPP) {
    x += dpp_row_shift<DPP_ROW_SHL + LSTEP, true>(x);      // sub <= 2: + level below
    x += dpp_row_shift<DPP_ROW_SHL + 2 * LSTEP, true>(x);  // sub <= 1: + the pair below
    incl = x;
    excl = dpp_row_shift<DPP_ROW_SHL + LSTEP, false>(x);   // sub == 3: 0.0
    return;
  }
  double t = shfl_abs(x, lane + LSTEP);
  if (sub <= 2) x += t;
  t = shfl_abs(x, lane + 2 * LSTEP);
  if (sub <= 1) x += t;
  incl = x;
  t = shfl_abs(x, lane + LSTEP);
  excl = sub <= 2 ? t : 0.0;
}

// Workgroup barrier.  __syncthreads() is a workgroup-scope fence + s_barrier: with global stores outstanding
// the fence waits for them too (s_waitcnt vmcnt(0)); in the persistent form that would stall every barrier on
// the previous element's stores, so there only the LDS counter is drained — the barriers order LDS traffic,
// all global data is wave-private.  (In the non-persistent kernels no store precedes a barrier and the compiled
// barriers are `s_waitcnt lgkmcnt(0); s_barrier`: loads requested before a barrier stay in flight across it.)
template <bool LDS_ONLY>
__device__ __forceinline__ void wg_barrier() {
  if constexpr (LDS_ONLY) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else __syncthreads();
}

// LDS image of the element's metric terms: 13 values per GLL point
enum { G_FCOR = 0, G_SPHEREMP = 16, G_METDET = 32, G_RMETDET = 48, G_PHIS = 64, G_D = 80, G_DINV = 144, G_SIZE = 208 };

// PERSIST: the workgroup walks elements blockIdx.x, blockIdx.x + gridDim.x, ... and requests the
// next element's n0 inputs while it computes the last phase of the current one, so neither the
// workgroup launch nor the first HBM round trip of an element is exposed (one workgroup per CU).
//
// VADV: the Eulerian vertical coordinate (rsplit == 0; routine_extracted.F90:224-262,515-517 = "X",
// level_vectorized_ppscan/CaarFunctor.hpp:505-547 preq_vertadv): the interface mass flux eta_dot_dpdn
// from the column total and the running sum of divdp, and the vertical advection of T and v, for
// which T, u, v of the whole column are mirrored in LDS (the level above / below a lane's own).
// No extra HBM traffic.  The reference never builds this branch: parity unpinned (oracle/caar_oracle.h).
//
// NLEV_T == 0: the level count is a run-time argument (k.nlev <= 4*DYNW*TPW): the workgroup has
// ceil(ceil(nlev/4)/TPW) waves, tiles and rows beyond the last level are dead (masked, see
// RAGGED), LDS is sized for the largest count.  Serves every PLEV the reference can be
// configured with (config.h.in:3) that has no kernel of its own.
// DYNW: the most waves a workgroup of that form may have (its launch bound).
// SNT: non-temporal (streaming) loads and stores; ANT: the same for the three read-modify-write
// accumulators (derived_vn0, omega_p, eta_dot_dpdn), which a hybrid-policy kernel keeps in the
// memory-side cache for part of the elements.
// The workgroup's LDS, declared ONCE in the kernel and shared by the code paths instantiated inside it (the
// hybrid cache policy compiles the element body twice; as function-local __shared__ arrays every buffer
// existed twice: 17.4 KB instead of 8.7 KB at NLEV=72).
template <int NLEV_T, int TPW, bool PERSIST, bool VADV, int DYNW, int PARK = 0, int CARRY_LDS = 0>
struct Np4Lds {
  static constexpr int PP = 16;
  static constexpr int NT_MAX = NLEV_T == 0 ? DYNW * TPW : (NLEV_T + 3) / 4;
  static constexpr int COL = VADV ? (NT_MAX * 4 + 2) * PP : 1;
  double dvv[16];
  double geo_buf[PERSIST ? 2 : 1][208];  // G_SIZE; double-buffered across elements
  // (CARRY_LDS, the step loops that carry state: two sets, used by alternate calls, so that a call's first phase may
  // overwrite tile totals while slower waves still read the previous call's in its last phase — no barrier between calls)
  static constexpr int TOTB = CARRY_LDS ? 2 : 1;
  double tot_dp[TOTB * NT_MAX * PP];     // sum of dp over each tile
  double tot_div[TOTB * NT_MAX * PP];    // sum of divdp over each tile
  double tot_ht[TOTB * NT_MAX * PP];     // sum of Rgas*T_v*dp/p over each tile
  // VADV: T, u, v at n0 of the whole column, [field][1 + level][pt] with a zero row above the top level and
  // below the bottom one (and room for the dead rows of a ragged last tile)
  double col[3][COL];
  double hybi[VADV ? NT_MAX * 4 + 1 : 1];
  // PARK (bit mask): p (1), the divdp prefix (2), the in-tile hydrostatic suffix (4), divdp (8), T_v (16) of every point
  // wait here between the phases ([slot][tile * 64 + lane]; each lane re-reads only what it wrote); 32 (Eulerian form):
  // u, v, T are re-read from `col` in the last phase instead of staying in registers
  static constexpr int NPARK = (PARK & 1) + ((PARK >> 1) & 1) + ((PARK >> 2) & 1) + ((PARK >> 3) & 1) + ((PARK >> 4) & 1);
  double park[NPARK ? NPARK : 1][NPARK ? NT_MAX * 64 : 1];
  // CARRY_LDS (step loop) >= 1: what the NEXT call will read as u, v, T, dp3d at nm1 (= this call's n0 state; slots 0-3) and
  // the tracer block (4), handed from call to call in LDS ([slot][tile * 64 + lane]; each lane re-reads only what it wrote);
  // 2: the accumulators as well — vn0 (5, 6), omega_p (7), eta_dot_dpdn (8) — and pecnd (9): a steady call then touches no
  // element array in memory at all; 3: of those only vn0 and omega_p (where LDS has no room for all five: NLEV=128)
  // 4: the nm1 state, vn0 and omega_p but NOT the tracer block (7 slots: with the NLEV=72 four-wave shape 80.2 KB per workgroup,
  // the most that leaves room for two workgroups per CU) — the tracer block, pecnd (and nothing else) are read per call
  static constexpr int NCARRY = CARRY_LDS == 2 ? 10 : (CARRY_LDS == 3 ? 8 : (CARRY_LDS == 4 ? 7 : (CARRY_LDS ? 5 : 1)));
  static constexpr int ACC0 = CARRY_LDS == 4 ? 4 : 5;  // first accumulator slot
  double carry[NCARRY][CARRY_LDS ? NT_MAX * 64 : 1];
};

// PARK: the five per-point values that live from the scans to the last phase (p, divdp prefix, hydrostatic in-tile suffix,
// divdp, T_v) wait in LDS instead of registers (10 VGPRs per tile): lets a fat shape (few waves x many tiles) stay within
// 256 registers, i.e. two workgroups per CU, where the level count makes the tiles-per-wave large (NLEV=128: 4 waves x 8)
// or the form holds more per level (Eulerian, NLEV=72: 4 waves x 5, 5, 4, 4 with p, divdp prefix, divdp parked and u, v, T re-read
// from the column copy).  The re-reads go through lds_reread_ptr (plain ds_read_b64): near free, 82.9 % against 83.6 % unparked at
// NLEV=72.  (Round 2 first had them as `volatile` generic loads = flat_load sc0 sc1 + s_waitcnt vmcnt(0), which drained the
// global loads in flight at every parked read: 75.4 %.)
// The n0 inputs of a wave's tiles (dp3d, u, v, T at time level n0; the tracer mass Qdp).  The step loop
// (caar_np4_steps_kernel) carries one of these from call to call: what a lane stores at np1 is what it would load as n0
// in the next call, so it keeps the values instead (CARRY).
template <int TPW>
struct Np4N0In {
  double dp[TPW], u[TPW], v[TPW], T[TPW], q[TPW];
};

// STEPS: the body runs inside the step loop of caar_np4_steps_kernel (several calls of the routine by one workgroup): the
// barriers drain the LDS counter only, like the persistent form's, so that a step's first barrier does not wait for the
// previous step's stores.
// CARRY_IN (compile time; STEPS only), what the previous call of the step loop handed over: bit 1 = `carry` holds this
// call's n0 state (dp3d, u, v, T), 2 = lds.carry holds its nm1 state, 4 = lds.carry holds its tracer block (2 and 4 only
// with CARRY_LDS), 8 = lds.carry holds the accumulators and pecnd, and the metric terms are still staged (CARRY_LDS == 2).
// Every STEPS call hands the same things on to the next one.
template <int NLEV_T, int TPW, int MINW, bool MOIST, bool SNT, bool ANT, int PF, bool PERSIST, bool ETA_COND, bool VADV, int DYNW, int PARK = 0, bool STEPS = false, int CARRY_LDS = 0, int CARRY_IN = 0, int STORES = -1, int WAVES = 0>
__device__ __forceinline__ void caar_np4_element(const KernelArgs& k, Np4Lds<NLEV_T, TPW, PERSIST, VADV, DYNW, PARK, CARRY_LDS>& lds,
                                                 Np4N0In<TPW>* carry = nullptr, int step_stores = 3, int tot_par = 0,
                                                 long long ie_known = -2) {
  // step_stores (uniform; STEPS only; else all; STORES >= 0: the same as a compile-time constant, for the hot loop):
  // bit 1 = store the np1 state (v, T, dp3d), bit 2 = store derived_phi.  The
  // step loop leaves them out where a later call of the same launch overwrites them and nothing reads them in between:
  // phi is never read (only the last call's survives); with the whole prognostic state carried on chip (CARRY_LDS) only
  // the last three calls' np1 states are what single launches would leave in the three time levels.
  static_assert(!CARRY_LDS || STEPS, "CARRY_LDS: step loop only");
  static_assert(CARRY_IN == 0 || STEPS, "CARRY_IN: step loop only");
  static_assert(CARRY_LDS || (CARRY_IN & 6) == 0, "nm1 / tracer carry needs CARRY_LDS");
  static_assert(CARRY_LDS != 4 || (CARRY_IN & 4) == 0, "CARRY_LDS == 4 does not carry the tracer block");
  static_assert(CARRY_LDS >= 2 || (CARRY_IN & 8) == 0, "accumulator carry needs CARRY_LDS >= 2");
  constexpr int carry_flags = CARRY_IN;
  constexpr bool carry_valid = STEPS && (CARRY_IN & 1);
  constexpr int PP = 16;               // GLL points per level
  constexpr bool DYN = NLEV_T == 0;
  constexpr bool SCAN_DPP = np4_scan_dpp<NLEV_T, STEPS>();
  constexpr int NT_MAX = DYN ? DYNW * TPW : (NLEV_T + 3) / 4;  // LDS sizing
  const int NLEV = DYN ? k.nlev : NLEV_T;
  // tiles per element incl. dead ones (the last live one partly empty if NLEV % 4 != 0)
  const int NT = DYN ? (int)(blockDim.x >> 6) * TPW : NT_MAX;
  constexpr bool RAGGED = DYN || NLEV_T % 4 != 0;
  // Tiles per wave.  Where TPW does not divide the tile count (UNEVEN; NLEV=72 as 4 waves: 5, 5, 4, 4 of its 18 tiles) the
  // first FULL waves own TPW consecutive tiles and the others TPW - 1: their last tile slot is dead, and everything that
  // is per tile sits under a wave-uniform branch (tile_live).  Why: two 3-wave workgroups per CU put 2, 2, 1, 1 waves on the
  // CU's four SIMDs and a kernel bound by instruction issue (the step loop) runs at the pace of the SIMD that holds two;
  // 4-wave workgroups land one wave per SIMD each (tools/probes/simd_placement_probe.hip, profiles/r03/simd_placement_probe.log).
  constexpr int WAVES_T = DYN ? DYNW : (WAVES ? WAVES : (NT_MAX + TPW - 1) / TPW);  // WAVES: more waves than ceil(tiles / TPW), e.g. 18 tiles as 3, 3, 2, 2, 2, 2, 2, 2
  constexpr int FULL = DYN ? DYNW : NT_MAX - WAVES_T * (TPW - 1);
  constexpr bool UNEVEN = !DYN && FULL != WAVES_T;
  const int THREADS = DYN ? (int)blockDim.x : WAVES_T * 64;
  const int BLK = NLEV * PP;           // doubles in one scalar field block
  static_assert(!WAVES || !DYN, "WAVES: compile-time level counts only");
  static_assert(DYN || (FULL >= 1 && FULL <= WAVES_T), "tile decomposition");
  static_assert(!UNEVEN || (TPW >= 2 && !PERSIST), "uneven tile counts: non-persistent form");
  static_assert(!RAGGED || !PERSIST, "ragged level counts: non-persistent form only");
  static_assert(!VADV || (!PERSIST && !ETA_COND), "Eulerian branch: plain form only");
  static_assert(PARK == 0 || !PERSIST, "PARK: non-persistent form");  // (ragged / dead rows park garbage nobody reads)

  static_assert(G_SIZE == 208, "Np4Lds::geo_buf");
  double* const s_dvv = lds.dvv;
  auto& s_geo_buf = lds.geo_buf;
  // tot_par (uniform; CARRY_LDS step loops): which of the two sets of tile totals this call uses
  double* const s_tot_dp = lds.tot_dp + (CARRY_LDS ? tot_par * (NT_MAX * PP) : 0);
  double* const s_tot_div = lds.tot_div + (CARRY_LDS ? tot_par * (NT_MAX * PP) : 0);
  double* const s_tot_ht = lds.tot_ht + (CARRY_LDS ? tot_par * (NT_MAX * PP) : 0);
  auto& s_col = lds.col;
  double* const s_hybi = lds.hybi;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pt = mfma4_point(lane);
  const int sub = mfma4_level(lane);
  const size_t tl = (size_t)k.timelevels;
  // RAGGED (NLEV not a multiple of 4): the rows of the last tile beyond level NLEV-1 are dead:
  // their loads are masked and return 0, they contribute 0 to the three integrals, and they
  // store nothing.  A level is one block of the 4x4x4 MFMA, so dead rows never feed live ones.
  const int tile0 = UNEVEN ? w * (TPW - 1) + (w < FULL ? w : FULL) : w * TPW;  // this wave's first tile (wave-uniform)
  const bool last_live = !UNEVEN || w < FULL;                                    // ... and whether it owns TPW tiles
  const int ntile = last_live ? TPW : TPW - 1;
  auto tile_live = [&](int r) { return !UNEVEN || r < TPW - 1 || last_live; };
  // Tile numbers, spelled for the even shapes exactly as before this option existed (through a named variable the
  // compiler schedules the headline kernels differently: measured 1 % slower)
#define CAAR_TILE0 (UNEVEN ? tile0 : w * TPW)
#define CAAR_TILE(r) (UNEVEN ? tile0 + (r) : w * TPW + (r))
#define CAAR_TILE_END (UNEVEN ? tile0 + ntile : w * TPW + TPW)
#define CAAR_LIVE(r) (UNEVEN ? (tile_live(r) && live_row(r)) : live_row(r))
  auto live_row = [&](int r) { return !RAGGED || (CAAR_TILE(r) * 4 + sub) < NLEV; };
  // Addressing: every field pointer below is wave-uniform (element, time level and this
  // wave's first tile folded in: an SGPR pair) and is indexed by `r * 64 + ulane` with r
  // a compile-time tile number and ulane an UNSIGNED lane id, so each access is one
  // global_load/store with scalar base, one shared 32-bit lane offset and an immediate.
  const unsigned ulane = sub * 16 + pt;  // this lane's offset inside a tile of the layout [lev][a][b]
  // (the even shapes keep the expression they always had: the headline kernels' code must not change with this option)
  const size_t wbase = UNEVEN ? (size_t)tile0 * 64 : (size_t)w * (TPW * 64);  // first point of this wave's tiles inside a field block

  // ie_known (uniform; >= -1): the caller has already mapped this workgroup to its element (the hybrid kernel needs it to
  // pick the code path, the step loop for its early exit): not formed a second time in front of the first loads
  long long ie_s = PERSIST ? (blockIdx.x < (unsigned)k.nelem ? (long long)k.nets + blockIdx.x : -1)
                           : (ie_known != -2 ? ie_known : element_of_block(k, blockIdx.x));
  if (ie_s < 0) return;  // padding block (uniform for the workgroup)
  unsigned eb = blockIdx.x;  // PERSIST: element counter relative to nets

  // n0 inputs of this wave's tiles
  using N0In = Np4N0In<TPW>;
  static_assert(!STEPS || (!PERSIST && !RAGGED && !VADV), "step loop: plain Lagrangian form");
  auto load_n0 = [&](size_t ie) {
    const double* __restrict__ dp_n0 = k.dp3d + (ie * tl + k.n0) * BLK + wbase;
    const dbl2* __restrict__ v_n0 = reinterpret_cast<const dbl2*>(k.v + (ie * tl + k.n0) * BLK * 2) + wbase;
    const double* __restrict__ T_n0 = k.T + (ie * tl + k.n0) * BLK + wbase;
    const double* __restrict__ Qdp = k.Qdp + ((ie * k.qsize_d + 0) * 2 + (MOIST ? k.qn0 : 0)) * BLK + wbase;
    N0In x;
#pragma unroll
    for (int r = 0; r < TPW; ++r) {
      x.dp[r] = x.u[r] = x.v[r] = x.T[r] = x.q[r] = 0.0;
      if (CAAR_LIVE(r)) {
        x.dp[r] = stream_load<SNT>(dp_n0 + r * 64 + ulane);
        const dbl2 uv = stream_load<SNT>(v_n0 + r * 64 + ulane);
        x.u[r] = uv.x;
        x.v[r] = uv.y;
        x.T[r] = stream_load<SNT>(T_n0 + r * 64 + ulane);
        x.q[r] = MOIST ? stream_load<SNT>(Qdp + r * 64 + ulane) : 0.0;
      }
    }
    return x;
  };

  // address of the idx-th entry of an element's LDS metric image (G_* layout)
  auto geo_src = [&](size_t ie, int idx) -> const double* {
    if (idx < G_SPHEREMP) return k.fcor + ie * PP + idx;
    if (idx < G_METDET) return k.spheremp + ie * PP + (idx - G_SPHEREMP);
    if (idx < G_RMETDET) return k.metdet + ie * PP + (idx - G_METDET);
    if (idx < G_PHIS) return k.rmetdet + ie * PP + (idx - G_RMETDET);
    if (idx < G_D) return k.phis + ie * PP + (idx - G_PHIS);
    if (idx < G_DINV) return k.D + ie * PP * 4 + (idx - G_D);
    return k.Dinv + ie * PP * 4 + (idx - G_DINV);
  };
  static_assert(!PERSIST || WAVES_T * 64 >= G_SIZE, "persistent form stages one metric value per thread");

  // ---- phase 0: issue the n0 loads of the first element --------------------------------
  // STEPS: `in` is the caller's carry.  carry_valid (uniform): it already holds this call's n0 state — the previous call's
  // np1 results, bit for bit what a load would return — so only the tracer block is requested.
  N0In in_local;
  N0In& in = STEPS ? *carry : in_local;
  if (STEPS && carry_valid) {
    if (CARRY_LDS && (carry_flags & 4)) {
      const lds_cptr cq = lds_reread_ptr(&lds.carry[4][0] + (UNEVEN ? (size_t)tile0 * 64 : (size_t)w * (TPW * 64)) + lane);
#pragma unroll
      for (int r = 0; r < TPW; ++r) {
        in.q[r] = 0.0;
        if (MOIST && tile_live(r)) in.q[r] = cq[r * 64];
      }
    } else {
      const double* __restrict__ Qdp = k.Qdp + (((size_t)ie_s * k.qsize_d + 0) * 2 + (MOIST ? k.qn0 : 0)) * BLK + wbase;
#pragma unroll
      for (int r = 0; r < TPW; ++r) {
        in.q[r] = 0.0;
        if (MOIST && tile_live(r)) in.q[r] = stream_load<SNT>(Qdp + r * 64 + ulane);
      }
    }
  } else {
    in = load_n0((size_t)ie_s);
  }
  if constexpr (CARRY_LDS && CARRY_LDS != 4 && MOIST) {
    if (!(carry_flags & 4)) {  // first call: the tracer block does not change from call to call (qn0 is fixed)
#pragma unroll
      for (int r = 0; r < TPW; ++r)
        if (tile_live(r)) lds.carry[4][CAAR_TILE(r) * 64 + lane] = in.q[r];
    }
  }
  double geo_reg = 0.0;
  if (PERSIST && tid < G_SIZE) geo_reg = *geo_src((size_t)ie_s, tid);
  if (tid < 16 && !(carry_flags & 1)) s_dvv[tid] = k.Dvv[tid];
  if (VADV) {
    for (int idx = tid; idx < NT * 4 + 1; idx += THREADS) s_hybi[idx] = idx <= NLEV ? k.hybi[idx] : 0.0;
    if (tid < PP) {
#pragma unroll
      for (int f = 0; f < 3; ++f) {
        s_col[f][tid] = 0.0;
        s_col[f][(NLEV + 1) * PP + tid] = 0.0;
      }
    }
  }
  Np4Ctx c;
  int par = 0;
  bool first = true;

  for (;;) {
    const size_t ie = (size_t)ie_s;
    double* const s_geo = s_geo_buf[PERSIST ? par : 0];
    double (&dp)[TPW] = in.dp;
    double (&u)[TPW] = in.u;
    double (&v)[TPW] = in.v;
    double (&T)[TPW] = in.T;
    double (&q)[TPW] = in.q;

    // pointers of the update phase
    const dbl2* __restrict__ v_nm1 = reinterpret_cast<const dbl2*>(k.v + (ie * tl + k.nm1) * BLK * 2) + wbase;
    const double* __restrict__ T_nm1 = k.T + (ie * tl + k.nm1) * BLK + wbase;
    const double* __restrict__ dp_nm1 = k.dp3d + (ie * tl + k.nm1) * BLK + wbase;
    dbl2* __restrict__ v_np1 = reinterpret_cast<dbl2*>(k.v + (ie * tl + k.np1) * BLK * 2) + wbase;
    double* __restrict__ T_np1 = k.T + (ie * tl + k.np1) * BLK + wbase;
    double* __restrict__ dp_np1 = k.dp3d + (ie * tl + k.np1) * BLK + wbase;
    dbl2* __restrict__ vn0 = reinterpret_cast<dbl2*>(k.vn0 + ie * BLK * 2) + wbase;
    double* __restrict__ omega_p = k.omega_p + ie * BLK + wbase;
    double* __restrict__ phi_out = k.phi + ie * BLK + wbase;
    const double* __restrict__ pecnd = k.pecnd + ie * BLK + wbase;
    double* __restrict__ eta = k.eta_dot_dpdn + ie * (BLK + PP) + wbase;
    double* __restrict__ eta_last = k.eta_dot_dpdn + ie * (BLK + PP) + BLK;

    // A later call of a steady step loop: the previous call already added eta_ave_w * 0 to this element's eta_dot_dpdn, and
    // x + 0 is a fixed point after one application (it only turns -0 into +0) — unless eta_ave_w * 0 is a NaN, and then
    // the host does not take the step loop (caar_abi.hip try_fused_steps).  The read-modify-write that changes nothing is
    // left out (CARRY_LDS == 2 carries the value in LDS instead).
    constexpr bool eta_rmw = VADV || CARRY_LDS == 2 || !(carry_flags & 1);
    // Update-phase inputs of one tile (nm1 state, vn0, omega_p, pecnd, eta).
    struct TileIn {
      dbl2 vnm1, vn0;
      double Tnm1, dpnm1, om, pec, eta;
    };
    auto load_tile = [&](int r) {
      const unsigned off = r * 64 + ulane;
      TileIn x = {};
      if (!CAAR_LIVE(r)) return x;
      if (CARRY_LDS && (carry_flags & 2)) {
        // the previous call parked its n0 state (this call's nm1) in LDS: it is read where it is used, not a tile ahead
        // (LDS latency needs no prefetch, and four values fewer are live per tile in flight)
      } else {
        x.vnm1 = stream_load<SNT>(v_nm1 + off);
        x.Tnm1 = stream_load<SNT>(T_nm1 + off);
        x.dpnm1 = stream_load<SNT>(dp_nm1 + off);
      }
      if (CARRY_LDS >= 2 && (carry_flags & 8)) {  // the previous call left the accumulators (2: and pecnd) in LDS
        const lds_cptr ca = lds_reread_ptr(&lds.carry[CARRY_LDS >= 2 ? std::remove_reference<decltype(lds)>::type::ACC0 : 0][0] + CAAR_TILE(r) * 64 + lane);
        constexpr int Q = NT_MAX * 64;
        x.vn0.x = ca[0];
        x.vn0.y = ca[Q];
        x.om = ca[2 * Q];
        if constexpr (CARRY_LDS == 2) {
          x.eta = ca[3 * Q];
          x.pec = ca[4 * Q];
        } else {
          x.pec = stream_load<SNT>(pecnd + off);
          x.eta = eta_rmw ? stream_load<ANT>(eta + off) : 0.0;
        }
      } else {
        x.vn0 = stream_load<ANT>(vn0 + off);
        x.om = stream_load<ANT>(omega_p + off);
        x.pec = stream_load<SNT>(pecnd + off);
        x.eta = eta_rmw ? stream_load<ANT>(eta + off) : 0.0;
      }
      return x;
    };
    // PF: when the update-phase inputs are requested: 2 = right behind the n0 loads,
    // 1 = all tiles before the last barrier, 0 = one tile ahead of their use.
    TileIn pre[PF ? TPW : 1];
    if (PF == 2) {
#pragma unroll
      for (int r = 0; r < TPW; ++r) pre[r] = load_tile(r);
    }

    // stage the element's metric terms in LDS (PERSIST: the value was requested during the
    // previous element's last phase, ahead of its stores)
    if (PERSIST) {
      if (tid < G_SIZE) s_geo[tid] = geo_reg;
    } else if (!(carry_flags & 1)) {  // (step loop, later call for the same element: the metric terms are still staged)
      for (int idx = tid; idx < G_SIZE; idx += THREADS) s_geo[idx] = stream_load<SNT>(geo_src(ie, idx));
    }
    // (a later call of the step loop staged nothing, and the loop's own barrier between the calls is the fence)
    if (!(carry_flags & 1)) wg_barrier<PERSIST || STEPS>();  // also fences the previous element's last reads of the tile totals

    if (first) {
      c = make_mfma4_ctx(s_dvv, lane);
      first = false;
    }
    M22 Dinv;
    Dinv.m00 = s_geo[G_DINV + pt * 4 + 0];
    Dinv.m01 = s_geo[G_DINV + pt * 4 + 1];
    Dinv.m10 = s_geo[G_DINV + pt * 4 + 2];
    Dinv.m11 = s_geo[G_DINV + pt * 4 + 3];
    const double metdet = s_geo[G_METDET + pt];
    const double rmetdet = s_geo[G_RMETDET + pt];
    const double rrearth = k.rrearth;

    // ---- phase 1: divdp, T_v, in-tile scans of dp and divdp ---------------------------
    double divdp[TPW], Tv[TPW], ex_dp[TPW], ex_div[TPW];
#pragma unroll
    for (int r = 0; r < TPW; ++r) {
      if constexpr (UNEVEN) {
        if (!tile_live(r)) continue;  // (wave-uniform)
      }
      const int t = CAAR_TILE(r);
      divdp[r] = divergence_sphere(c, Dinv, metdet, rmetdet, rrearth, u[r] * dp[r], v[r] * dp[r]);  // P:114-121
      Tv[r] = MOIST ? T[r] * (1.0 + k.rv_over_rd_m1 * (q[r] * recip(dp[r]))) : T[r];               // P:135,150-151
      if (RAGGED && !live_row(r)) Tv[r] = 0.0;  // dead row: dp == 0 made the line above NaN
      double in_dp, in_div;
      scan_down<SCAN_DPP>(dp[r], lane, sub, in_dp, ex_dp[r]);
      scan_down<SCAN_DPP>(divdp[r], lane, sub, in_div, ex_div[r]);
      if (sub == 3) {
        s_tot_dp[t * PP + pt] = in_dp;
        s_tot_div[t * PP + pt] = in_div;
      }
      if (VADV && live_row(r)) {
        s_col[0][PP + t * 64 + ulane] = T[r];
        s_col[1][PP + t * 64 + ulane] = u[r];
        s_col[2][PP + t * 64 + ulane] = v[r];
      }
    }
    wg_barrier<PERSIST || STEPS>();

    // ---- phase 2: p, running divdp sum, hydrostatic increments and their in-tile scan --
    double p[TPW], rp[TPW], suml[TPW], ex_ht[TPW];
    double sdot_sum = 0.0;  // VADV: column total of divdp (X:237)
    {
      double base_dp = 0.0, base_div = 0.0;
      for (int t2 = 0; t2 < CAAR_TILE0; ++t2) {  // tiles above this wave's first tile (wave-uniform trip count)
        base_dp += s_tot_dp[t2 * PP + pt];
        base_div += s_tot_div[t2 * PP + pt];
      }
#pragma unroll
      for (int r = 0; r < TPW; ++r) {
        if constexpr (UNEVEN) {
          if (!tile_live(r)) continue;
        }
        const int t = CAAR_TILE(r);
        p[r] = (k.p_top + (base_dp + ex_dp[r])) + 0.5 * dp[r];  // P:84,94-96 in closed form
        suml[r] = base_div + ex_div[r];                          // P:327,339: sum of divdp above
        rp[r] = recip(p[r]);
        double ht = (k.Rgas * Tv[r]) * (dp[r] * rp[r]);          // Rgas*T_v*hkl, hkl = dp/p (P:300-302)
        if (RAGGED && !live_row(r)) ht = 0.0;
        double in_ht;
        scan_up<SCAN_DPP>(ht, lane, sub, in_ht, ex_ht[r]);
        if (sub == 0) s_tot_ht[t * PP + pt] = in_ht;
        base_dp += s_tot_dp[t * PP + pt];
        base_div += s_tot_div[t * PP + pt];
        if constexpr (PARK != 0) {
          constexpr int S1 = PARK & 1, S2 = S1 + ((PARK >> 1) & 1), S3 = S2 + ((PARK >> 2) & 1), S4 = S3 + ((PARK >> 3) & 1);
          if constexpr (PARK & 1) lds.park[0][t * 64 + lane] = p[r];
          if constexpr (PARK & 2) lds.park[S1][t * 64 + lane] = suml[r];
          if constexpr (PARK & 4) lds.park[S2][t * 64 + lane] = ex_ht[r];
          if constexpr (PARK & 8) lds.park[S3][t * 64 + lane] = divdp[r];
          if constexpr (PARK & 16) lds.park[S4][t * 64 + lane] = Tv[r];
        }
      }
      if (VADV) {
        for (int t2 = CAAR_TILE_END; t2 < NT; ++t2) base_div += s_tot_div[t2 * PP + pt];
        sdot_sum = base_div;
      }
    }

    if (PF == 1) {
#pragma unroll
      for (int r = 0; r < TPW; ++r) pre[r] = load_tile(r);
    }
    TileIn cur = PF ? pre[TPW - 1] : load_tile(TPW - 1);  // in flight across the barrier
    double l_eta_last = 0.0;
    if (tid < PP && eta_rmw) l_eta_last = eta_last[(unsigned)tid];
    wg_barrier<PERSIST || STEPS>();

    // PERSIST: request the next element's n0 inputs now; they land while phase 3 computes
    N0In nxt_in;
    long long nxt_ie = -1;
    if (PERSIST) {
      const unsigned eb_next = eb + gridDim.x;
      if (eb_next < (unsigned)k.nelem) {
        nxt_ie = (long long)k.nets + eb_next;
        nxt_in = load_n0((size_t)nxt_ie);
        if (tid < G_SIZE) geo_reg = *geo_src((size_t)nxt_ie, tid);
      }
      eb = eb_next;
    }

    // ---- phase 3: everything else, level-local -------------------------------------------
    M22 Dm;
    Dm.m00 = s_geo[G_D + pt * 4 + 0];
    Dm.m01 = s_geo[G_D + pt * 4 + 1];
    Dm.m10 = s_geo[G_D + pt * 4 + 2];
    Dm.m11 = s_geo[G_D + pt * 4 + 3];
    const double fcor = s_geo[G_FCOR + pt];
    const double spheremp = s_geo[G_SPHEREMP + pt];
    const double phis = s_geo[G_PHIS + pt];
    const double eta_zero = k.eta_ave_w * 0.0;  // eta_dot_dpdn_tmp == 0 (P:22,172): vertically Lagrangian

    double below = 0.0;  // hydrostatic sum over the tiles below this wave's last tile
    for (int t2 = NT - 1; t2 > CAAR_TILE_END - 1; --t2) below += s_tot_ht[t2 * PP + pt];

#pragma unroll
    for (int rr = 0; rr < TPW; ++rr) {
      const int r = TPW - 1 - rr;  // bottom tile of the wave first: `below` accumulates upwards
      const int t = CAAR_TILE(r);
      const unsigned off = r * 64 + ulane;
      TileIn nxt = cur;
      if (r > 0) nxt = PF ? pre[r - 1] : load_tile(r - 1);
      if constexpr (UNEVEN) {
        if (!tile_live(r)) {  // (wave-uniform; only r == TPW - 1)
          cur = nxt;
          continue;
        }
      }
      double p_r, suml_r, exht_r, divdp_r, Tv_r;  // this tile's scan results (registers, or parked in LDS)
      if constexpr (PARK == 0) {
        p_r = p[r];
        suml_r = suml[r];
        exht_r = ex_ht[r];
        divdp_r = divdp[r];
        Tv_r = Tv[r];
      } else {
        const lds_cptr pk = lds_reread_ptr(&lds.park[0][0] + t * 64 + lane);  // must not be forwarded through registers
        constexpr int Q = NT_MAX * 64;
        constexpr int S1 = PARK & 1, S2 = S1 + ((PARK >> 1) & 1), S3 = S2 + ((PARK >> 2) & 1), S4 = S3 + ((PARK >> 3) & 1);
        if constexpr (PARK & 1) p_r = pk[0]; else p_r = p[r];
        if constexpr (PARK & 2) suml_r = pk[Q * S1]; else suml_r = suml[r];
        if constexpr (PARK & 4) exht_r = pk[Q * S2]; else exht_r = ex_ht[r];
        if constexpr (PARK & 8) divdp_r = pk[Q * S3]; else divdp_r = divdp[r];
        if constexpr (PARK & 16) Tv_r = pk[Q * S4]; else Tv_r = Tv[r];
      }
      double rp_r;
      if constexpr (PARK & 1) rp_r = recip(p_r); else rp_r = rp[r];
      // PARK bit 32 (Eulerian form only): u, v, T of this tile are re-read from the column copy phase 1 left in LDS
      double u_r, v_r, T_r;
      if constexpr (VADV && (PARK & 32)) {
        const lds_cptr cc = lds_reread_ptr(&s_col[0][0] + PP + t * 64 + ulane);
        constexpr int CS = sizeof(s_col[0]) / sizeof(double);
        T_r = cc[0];
        u_r = cc[CS];
        v_r = cc[2 * CS];
      } else {
        u_r = u[r];
        v_r = v[r];
        T_r = T[r];
      }

      const double ht = (k.Rgas * Tv_r) * (dp[r] * rp_r);         // same expression as in phase 2
      const double phi = (phis + (below + exht_r)) + 0.5 * ht;    // P:303,309
      below += s_tot_ht[t * PP + pt];

      double gp0, gp1;
      gradient_sphere(c, Dinv, rrearth, p_r, gp0, gp1);            // P:103
      const double vgrad_p = dot2(u_r, gp0, v_r, gp1);            // P:111
      const double ckk = 0.5 * rp_r, ckl = rp_r;                  // P:333-334 (ckl = 2*ckk)
      const double om = __builtin_fma(-ckk, divdp_r, __builtin_fma(vgrad_p, rp_r, -(ckl * suml_r)));  // P:325,336,348
      const double vort = vorticity_sphere(c, Dm, rmetdet, rrearth, u_r, v_r);  // P:122

      const double Ephi = 0.5 * dot2(u_r, u_r, v_r, v_r) + phi + cur.pec;  // P:196
      double gT0, gT1, gE0, gE1;
      gradient_sphere(c, Dinv, rrearth, T_r, gT0, gT1);            // P:200
      const double vgrad_T = dot2(u_r, gT0, v_r, gT1);            // P:209
      gradient_sphere(c, Dinv, rrearth, Ephi, gE0, gE1);            // P:213
      const double gpterm = Tv_r * rp_r;                          // P:219
      const double glnps1 = k.Rgas * gpterm * gp0;                  // P:221
      const double glnps2 = k.Rgas * gpterm * gp1;                  // P:222
      double vtens1 = v_r * (fcor + vort) - gE0 - glnps1;          // P:227 (v_vadv == 0)
      double vtens2 = -u_r * (fcor + vort) - gE1 - glnps2;         // P:228
      double ttens = -vgrad_T + k.kappa * Tv_r * om;               // P:230 (T_vadv == 0)
      double eta_lo = 0.0, eta_hi = 0.0;  // interface mass flux above / below this level
      if (VADV) {
        const int lev = t * 4 + sub;
        // X:238-254: eta_dot(k+1) = hybi(k+1)*sdot_sum - sum_{l<=k} divdp(l); 0 at the top and the surface
        eta_lo = lev == 0 ? 0.0 : s_hybi[lev] * sdot_sum - suml_r;
        eta_hi = lev >= NLEV - 1 ? 0.0 : s_hybi[lev + 1] * sdot_sum - (suml_r + divdp_r);
        const double half_rdp = 0.5 * recip(dp[r]);                 // X:118
        const double facp = half_rdp * eta_hi, facm = half_rdp * eta_lo;   // CaarFunctor.hpp:526-527
        const int ci = PP + t * 64 + ulane;
        // CaarFunctor.hpp:513-546 (the zero rows of s_col stand in for the missing one-sided terms)
        const double T_vadv = facp * (s_col[0][ci + PP] - T_r) + facm * (T_r - s_col[0][ci - PP]);
        const double u_vadv = facp * (s_col[1][ci + PP] - u_r) + facm * (u_r - s_col[1][ci - PP]);
        const double v_vadv = facp * (s_col[2][ci + PP] - v_r) + facm * (v_r - s_col[2][ci - PP]);
        vtens1 = -u_vadv + v_r * (fcor + vort) - gE0 - glnps1;     // X:326-328
        vtens2 = -v_vadv - u_r * (fcor + vort) - gE1 - glnps2;     // X:332-334
        ttens = -T_vadv - vgrad_T + k.kappa * Tv_r * om;           // X:338
      }

      if (RAGGED && !live_row(r)) {
        cur = nxt;
        continue;
      }
      dbl2 vnm1 = cur.vnm1;
      double Tnm1 = cur.Tnm1, dpnm1 = cur.dpnm1;
      if constexpr (CARRY_LDS && (carry_flags & 2)) {
        const lds_cptr cn = lds_reread_ptr(&lds.carry[0][0] + t * 64 + lane);
        constexpr int Q = NT_MAX * 64;
        vnm1.x = cn[0];
        vnm1.y = cn[Q];
        Tnm1 = cn[2 * Q];
        dpnm1 = cn[3 * Q];
      }
      dbl2 vo;
      vo.x = spheremp * (vnm1.x + k.dt2 * vtens1);                  // P:251
      vo.y = spheremp * (vnm1.y + k.dt2 * vtens2);                  // P:252
      const int st_mask = STORES >= 0 ? STORES : step_stores;
      const bool st_state = !STEPS || (st_mask & 1), st_phi = !STEPS || (st_mask & 2);
      const bool st_acc = CARRY_LDS < 2 || (st_mask & 4);  // accumulators carried in LDS: only the last call's reach memory
      const bool st_eta = CARRY_LDS != 2 ? true : st_acc;    // (3: eta_dot_dpdn is not carried)
      if (st_state) stream_store<SNT>(v_np1 + off, vo);
      const double T_new = spheremp * (Tnm1 + k.dt2 * ttens);                         // P:253
      if (st_state) stream_store<SNT>(T_np1 + off, T_new);
      const double dp_new = VADV ? spheremp * (dpnm1 - k.dt2 * (divdp_r + eta_hi - eta_lo))   // X:515-517
                                 : spheremp * (dpnm1 - k.dt2 * divdp_r);                       // P:254
      debug_check_dp3d(dp_new);  // CaarFunctor.hpp:82-97 check_dp3d (-DCAAR_DEBUG builds only)
      if (st_state) stream_store<SNT>(dp_np1 + off, dp_new);
      if (st_phi) stream_store<SNT>(phi_out + off, phi);                            // P:294,303,309
      const double om_new = cur.om + k.eta_ave_w * om;                              // P:173
      if (st_acc) stream_store<ANT>(omega_p + off, om_new);
      dbl2 vn;
      vn.x = cur.vn0.x + k.eta_ave_w * (u_r * dp[r]);               // P:117
      vn.y = cur.vn0.y + k.eta_ave_w * (v_r * dp[r]);               // P:118
      if (st_acc) stream_store<ANT>(vn0 + off, vn);
      {
        const double e_new = cur.eta + (VADV ? k.eta_ave_w * eta_lo : eta_zero);  // P:172, X:271-272
        // ETA_COND: the update adds eta_ave_w * 0 (vertically Lagrangian), so the stored value
        // differs from the loaded one only for -0.0 or a non-finite eta_ave_w; storing only
        // then keeps the array bit-identical to the reference's and drops the write traffic.
        if ((!ETA_COND || __double_as_longlong(e_new) != __double_as_longlong(cur.eta)) && st_eta && eta_rmw)
          stream_store<ANT>(eta + off, e_new);
        if constexpr (CARRY_LDS >= 2) {
          constexpr int A = CARRY_LDS >= 2 ? std::remove_reference<decltype(lds)>::type::ACC0 : 0;
          lds.carry[A][t * 64 + lane] = vn.x;
          lds.carry[A + 1][t * 64 + lane] = vn.y;
          lds.carry[A + 2][t * 64 + lane] = om_new;
          if constexpr (CARRY_LDS == 2) {
            lds.carry[A + 3][t * 64 + lane] = e_new;
            if (!(carry_flags & 8)) lds.carry[A + 4][t * 64 + lane] = cur.pec;  // (first call; pecnd never changes)
          }
        }
      }
      if constexpr (CARRY_LDS) {  // this call's n0 state is the next call's nm1
        lds.carry[0][t * 64 + lane] = u_r;
        lds.carry[1][t * 64 + lane] = v_r;
        lds.carry[2][t * 64 + lane] = T_r;
        lds.carry[3][t * 64 + lane] = dp[r];
      }
      if constexpr (STEPS) {  // the state just stored at np1 is the next call's n0 (update_time_levels): keep it
        // (through an opaque move: the next call must treat these exactly like loaded values — with their producing
        // multiplies in sight the compiler contracts the next call's first operations differently, a last-bit difference
        // from the single launches)
        double cu = vo.x, cv = vo.y, cT = T_new, cdp = dp_new;
        asm volatile("" : "+v"(cu), "+v"(cv), "+v"(cT), "+v"(cdp));
        in.u[r] = cu;
        in.v[r] = cv;
        in.T[r] = cT;
        in.dp[r] = cdp;
      }
      cur = nxt;
    }
    if (tid < PP && eta_rmw) {
      const double e_new = l_eta_last + eta_zero;                     // P:181
      if (!ETA_COND || __double_as_longlong(e_new) != __double_as_longlong(l_eta_last)) eta_last[(unsigned)tid] = e_new;
    }

    if (!PERSIST || nxt_ie < 0) break;
    in = nxt_in;
    ie_s = nxt_ie;
    par ^= 1;
  }
}

#undef CAAR_TILE0
#undef CAAR_TILE
#undef CAAR_TILE_END
#undef CAAR_LIVE

// The kernel.  POL = cache policy of the element arrays' loads and stores:
//   1  non-temporal: every array is touched once per launch, nothing is worth keeping (+0.4..5 % over 0);
//   0  the default policy;
//   2  hybrid: streaming like 1, except that for k.cache_count evenly spread elements the three
//      read-modify-write accumulators (vn0, omega_p, eta_dot_dpdn: read AND written by every call, so
//      a byte of them kept on chip saves two HBM transfers per call) use the default policy.
//      Streaming traffic does not allocate in the memory-side Infinity Cache (256 MB), so those blocks
//      survive the rest of the launch, and the next call on the same arrays (a time-stepping host, the
//      reference's driver loop) finds them there instead of in HBM.  Both code paths live in the
//      kernel; the choice is uniform per workgroup.
template <int NLEV_T, int TPW, int MINW, bool MOIST, int POL, int PF, bool PERSIST, bool ETA_COND, bool VADV = false, int DYNW = 8, int PARK = 0>
__global__ __launch_bounds__(NLEV_T ? ((NLEV_T + 3) / 4 + TPW - 1) / TPW * 64 : DYNW * 64, MINW) void caar_np4_kernel(const KernelArgs k) {
  request_kernel_args(k);
  __shared__ Np4Lds<NLEV_T, TPW, PERSIST, VADV, DYNW, PARK> lds;
  if constexpr (POL == 2) {
    static_assert(!PERSIST, "hybrid cache policy: non-persistent form only");
    const long long ie_s = element_of_block(k, blockIdx.x);
    if (ie_s < 0) return;
    if (element_is_cached(k, ie_s))
      caar_np4_element<NLEV_T, TPW, MINW, MOIST, true, false, PF, PERSIST, ETA_COND, VADV, DYNW, PARK>(k, lds, nullptr, 3, 0, ie_s);
    else
      caar_np4_element<NLEV_T, TPW, MINW, MOIST, true, true, PF, PERSIST, ETA_COND, VADV, DYNW, PARK>(k, lds, nullptr, 3, 0, ie_s);
  } else {
    caar_np4_element<NLEV_T, TPW, MINW, MOIST, POL == 1, POL == 1, PF, PERSIST, ETA_COND, VADV, DYNW, PARK>(k, lds);
  }
}

}  // namespace caar
