// caar_np8.hip — compute_and_apply_rhs for NP=8 on gfx950 (MI355X), hand-written HIP.
//
// Same algorithm and reference citations as caar_np4.hip (P = cxx/pointers_only/
// compute_and_apply_rhs.cpp, S = sphere_operators.cpp); different mapping because one
// level of an NP=8 element is exactly one wavefront:
//   * one workgroup = one element; lane -> GLL point by the MFMA result layout (np8::mfma_point; a*8+b in the
//     LDS-tile comparator variant); a wave owns TPW consecutive
//     levels and walks them in registers, so the three vertical integrals are plain
//     running sums inside a wave (the reference's own summation order within the
//     chunk) plus one wave total per integral exchanged through LDS (two barriers).
//   * every field access of a wave is one contiguous 512 B (scalar) / 1 KiB (v) row of
//     the reference layout [lev][a][b]: each byte read once, written once.
//   * the 8x8 Dvv contractions run on the matrix cores (DEFAULT, MFMA = true): v_mfma_f64_4x4x4, two issues per
//     8x8 product, operands placed by one bank-masked DPP row_ror (d/da) or one ds_bpermute (d/db) per k-block,
//     results landing where the pointwise code needs them — no LDS tile, no LDS copy of Dvv (caar_np8_ops.h "MFMA
//     form"; +1 % A/B against the direct form, profiles/r02/kbench_np8_nlev72_mfma.log).  The direct form (one
//     comparator variant): a wave-private 512 B LDS tile, the field written once and each lane reading its row
//     (4 x ds_read_b128) or column (8 x ds_read_b64), Dvv staged transposed in LDS.
//   * an NP=8 element does not fit the register file the way an NP=4 one does (7 live
//     values x 64 points x 72 levels = 258 KB): between the phases dp, u, v, T of every
//     level are read through a two-deep prefetch ring, dp, u, v are parked in LDS
//     (3 x 36 KB, each lane re-reads only what it wrote) and only {T, T_v, divdp}
//     stay in registers; p, 1/p and the divdp prefix are re-formed
//     in the last phase from the running sums.
#include <hip/hip_runtime.h>

#include "caar_kernel_args.h"
#include "caar_np8_ops.h"

namespace caar {


//
// VADV: the Eulerian vertical coordinate (rsplit == 0), see caar_np4.hip.  u and v of the
// neighbouring levels are already in the LDS park; T of the level above a wave's first and
// below its last level goes through a small LDS halo.
//
// MFMA: the 8x8 contractions go through v_mfma_f64_4x4x4 (caar_np8_ops.h "MFMA form"): the lane -> GLL point
// mapping becomes the MFMA result layout (mfma_point), the wave-private LDS tile and the LDS Dvv copy are not
// used at all, everything else is unchanged.
// LA: how many levels ahead the update-phase inputs (nm1 state, vn0, omega_p, pecnd, eta) are requested.
// The workgroup's LDS, declared ONCE in each kernel and shared by the instantiations of the body inside it (the step loop has
// two: as function-local __shared__ arrays every buffer would exist twice, 260 KB).
// DB (the step loop): two sets of wave totals, used by alternate calls — a call's first phase may then overwrite totals
// while slower waves still read the previous call's in its last phase, and the calls need no barrier between them.
template <int NLEV, int TPW, bool BATCH, bool VADV, bool MFMA, bool DB = false>
struct Np8Lds {
  static constexpr int WAVES = NLEV / TPW, BLK = NLEV * np8::PP, SLOTS = BATCH ? 5 : 1;
  __attribute__((aligned(16))) double dvvT[64];
  __attribute__((aligned(16))) double geo[np8::G_SIZE];
  __attribute__((aligned(16))) double tile[MFMA ? 1 : WAVES * 64 * SLOTS];  // LDS tile slots per wave (BATCH: p, T, Ephi, vcov1, vcov0)
  double park[3 * BLK + (VADV ? np8::PP : 0)];  // dp, u, v of every level, [field][lev][pt] (+ a zero row for VADV)
  double tot_dp[(DB ? 2 : 1) * WAVES * np8::PP];   // per wave: sum of dp over its levels
  double tot_div[(DB ? 2 : 1) * WAVES * np8::PP];  // ... of divdp
  double tot_ht[(DB ? 2 : 1) * WAVES * np8::PP];   // ... of Rgas*T_v*dp/p
  double Thalo[VADV ? WAVES * 2 * np8::PP : 1]; // VADV: T of each wave's first / last level
};

// STEPS: the body runs inside the step loop of caar_np8_steps_kernel (see caar_np4_steps.hip for the idea): barriers drain
// the LDS counter only, and the call hands its np1 results to the next one — dp3d, u, v through the LDS park they already
// live in, T through `cy->T` (registers) — and keeps the n0 state it replaces there in `cy->m*` (registers): with rotating
// time levels that is the next call's nm1 state.  CARRY_IN (compile time) 1: this call's n0 state arrives that way instead of
// being loaded (every lane reads and writes only its own points, so what it stored is what it would load); 2: its nm1 state
// too.  store_state = false: the np1 state is not written (a later call of the loop overwrites that time level and nothing
// reads it from memory before).
template <int TPW>
struct Np8Carry {
  double T[TPW];                               // T at n0 of this wave's levels
  double mdp[TPW], mu[TPW], mv[TPW], mT[TPW];  // the state one call back (nm1 after the rotation)
};
template <int NLEV, int TPW, int MINW, bool MOIST, bool SNT, bool COEF_LDS, bool RELOAD_T, bool BATCH, bool VADV = false, bool MFMA = false, int LA = 1,
          bool STEPS = false, int CARRY_IN = 0>
__device__ __forceinline__ void caar_np8_element(const KernelArgs& k, Np8Lds<NLEV, TPW, BATCH, VADV, MFMA, STEPS>& lds, Np8Carry<TPW>* cy = nullptr,
                                                 bool store_phi = true /* STEPS: false where a later call overwrites it */,
                                                 bool store_state = true, int tot_par = 0 /* STEPS: which set of wave totals */) {
  using namespace np8;
  static_assert(!STEPS || (!VADV && !RELOAD_T && !BATCH), "step loop: plain Lagrangian form");
  static_assert(!CARRY_IN || STEPS, "CARRY_IN: step loop only");
  constexpr bool CARRY_M = CARRY_IN == 2;
  auto wg_sync = [] {
    if constexpr (STEPS) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // not the previous call's stores
    else __syncthreads();
  };
  constexpr int WAVES = NLEV / TPW;
  constexpr int THREADS = WAVES * 64;
  constexpr int BLK = NLEV * PP;
  static_assert(NLEV % TPW == 0 && THREADS <= 1024, "level decomposition");
  static_assert(!VADV || !RELOAD_T, "Eulerian branch keeps T in registers");
  static_assert(!MFMA || (!BATCH && !COEF_LDS), "MFMA form: no LDS tile, Dvv slices are per-lane MFMA operands");

  constexpr int SLOTS = BATCH ? 5 : 1;
  double* const s_dvvT = lds.dvvT;
  double* const s_geo = lds.geo;
  double* const s_tile = lds.tile;
  double* const s_park = lds.park;
  double* const s_tot_dp = lds.tot_dp + (STEPS ? tot_par * (NLEV / TPW * np8::PP) : 0);
  double* const s_tot_div = lds.tot_div + (STEPS ? tot_par * (NLEV / TPW * np8::PP) : 0);
  double* const s_tot_ht = lds.tot_ht + (STEPS ? tot_par * (NLEV / TPW * np8::PP) : 0);
  double* const s_Thalo = lds.Thalo;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pt = MFMA ? mfma_point(lane) : lane;  // GLL point a*8+b of this lane
  // Slot of this lane's point inside the per-point LDS tables (metric terms, wave totals, T halo): the lane itself — the
  // MFMA form stages the tables permuted (mfma_lane_of_point), so that every table read of a wave covers 64 consecutive
  // doubles (indexed by `pt` the reads were 2-way bank conflicts, caar_np8_ops.h load_m22).
  const int sl = lane;
  const long long ie_s = element_of_block(k, blockIdx.x);
  if (ie_s < 0) return;  // padding block of the XCD-chunked grid (uniform for the workgroup)
  const size_t ie = (size_t)ie_s;
  const size_t tl = (size_t)k.timelevels;
  const int lev0 = w * TPW;
  // Addressing as in caar_np4.hip: wave-uniform field pointers (element, time level and
  // this wave's first level folded in) indexed by `r * PP + ulane`, r compile-time.
  const unsigned ulane = pt;  // offset of this lane's point inside a level
  const size_t wbase = (size_t)lev0 * PP;

  const double* __restrict__ dp_n0 = k.dp3d + (ie * tl + k.n0) * BLK + wbase;
  const dbl2* __restrict__ v_n0 = reinterpret_cast<const dbl2*>(k.v + (ie * tl + k.n0) * BLK * 2) + wbase;
  const double* __restrict__ T_n0 = k.T + (ie * tl + k.n0) * BLK + wbase;
  const double* __restrict__ Qdp = k.Qdp + ((ie * k.qsize_d + 0) * 2 + (MOIST ? k.qn0 : 0)) * BLK + wbase;

  // ---- phase 0: start the n0 loads; stage Dvv^T and the metric terms in LDS ----------
  struct N0In {
    double dp, T, q;
    dbl2 uv;
  };
  auto load_n0 = [&](int r) {
    const unsigned off = r * PP + ulane;
    N0In x;
    if constexpr (CARRY_IN) {  // dp, u, v: where the previous call left them (read in phase 1); T: its registers
      x.dp = 0.0;
      x.uv = dbl2{0.0, 0.0};
      x.T = 0.0;
    } else {
      x.dp = stream_load<SNT>(dp_n0 + off);
      x.uv = stream_load<SNT>(v_n0 + off);
      x.T = stream_load<SNT && !RELOAD_T>(T_n0 + off);  // RELOAD_T: default policy, re-read from L2 in phase 3
    }
    x.q = MOIST ? stream_load<SNT>(Qdp + off) : 0.0;
    return x;
  };
  constexpr int PD = TPW < 3 ? TPW : 3;  // levels in flight per wave
  N0In ring[PD];
#pragma unroll
  for (int r = 0; r < PD; ++r) ring[r] = load_n0(r);
  double T_local[RELOAD_T ? 1 : TPW], Tv[TPW];
  double* const T = STEPS ? cy->T : T_local;
  double* const park_dp = s_park + lev0 * PP + lane;  // + r*PP; u, v follow at BLK strides
  // re-reads through a laundered LDS pointer: the compiler must not forward the parked values through registers
  const lds_cptr park_rd = lds_reread_ptr(park_dp);
  if (tid < 64 && !CARRY_IN) s_dvvT[(tid & 7) * NP + (tid >> 3)] = k.Dvv[tid];  // Dvv[k][j] -> dvvT[j][k]
  if (VADV && tid < PP) s_park[3 * BLK + tid] = 0.0;
  // (a later call of the step loop works on the same element: its metric terms are still staged, and the loop's barrier
  // between the calls is the fence this barrier would be)
  for (int idx = tid; idx < (CARRY_IN ? 0 : G_SIZE); idx += THREADS) {
    const double* src;
    if (idx < G_SPHEREMP) src = k.fcor + ie * PP + idx;
    else if (idx < G_METDET) src = k.spheremp + ie * PP + (idx - G_SPHEREMP);
    else if (idx < G_RMETDET) src = k.metdet + ie * PP + (idx - G_METDET);
    else if (idx < G_PHIS) src = k.rmetdet + ie * PP + (idx - G_RMETDET);
    else if (idx < G_D) src = k.phis + ie * PP + (idx - G_PHIS);
    else if (idx < G_DINV) src = k.D + ie * PP * 4 + (idx - G_D);
    else src = k.Dinv + ie * PP * 4 + (idx - G_DINV);
    // scalars: [point] -> [slot]; D, Dinv: [point][r][c] in memory -> [r*2 + c][slot] in LDS (load_m22)
    int d;
    if (idx >= G_D) {
      const int base = idx < G_DINV ? G_D : G_DINV, i = idx - base;
      d = base + (i & 3) * PP + (MFMA ? mfma_lane_of_point(i >> 2) : (i >> 2));
    } else {
      d = (idx & ~(PP - 1)) + (MFMA ? mfma_lane_of_point(idx & (PP - 1)) : (idx & (PP - 1)));
    }
    s_geo[d] = stream_load<SNT>(src);
  }
  if (!CARRY_IN) wg_sync();

  MfmaCtx mc;
  if (MFMA) mc = make_mfma_ctx(k.Dvv, lane);
  Ctx c;
  c.tile = s_tile + (MFMA ? 0 : w * 64 * SLOTS);
  c.a = lane >> 3;
  c.b = lane & 7;
  c.dvvT = s_dvvT;
  if (!COEF_LDS) {
#pragma unroll
    for (int kk = 0; kk < NP; ++kk) {
      c.ca[kk] = s_dvvT[c.a * NP + kk];
      c.cb[kk] = s_dvvT[c.b * NP + kk];
    }
  }
  const double rrearth = k.rrearth;
  const double rmetdet = s_geo[G_RMETDET + sl];

  // ---- phase 1: divdp, T_v; wave totals of dp and divdp -------------------------------
  double divdp[TPW];
  {
    const M22 Dinv = load_m22(s_geo + G_DINV, sl);
    const double metdet = s_geo[G_METDET + sl];
    double run_dp = 0.0, run_div = 0.0;
#pragma unroll
    for (int r = 0; r < TPW; ++r) {
      N0In x = ring[r % PD];
      if (r + PD < TPW) ring[r % PD] = load_n0(r + PD);
      if constexpr (CARRY_IN) {
        x.dp = park_rd[r * PP];
        x.uv.x = park_rd[BLK + r * PP];
        x.uv.y = park_rd[2 * BLK + r * PP];
        x.T = T[r];
      }
      if (BATCH) {  // both contravariant components go to their slots, one LDS round trip
        const double vdp0 = x.uv.x * x.dp, vdp1 = x.uv.y * x.dp;                     // P:114-115
        const double gv0 = metdet * (Dinv.m00 * vdp0 + Dinv.m01 * vdp1);             // S:66-67
        const double gv1 = metdet * (Dinv.m10 * vdp0 + Dinv.m11 * vdp1);             // S:68-69
        wave_lds_fence();
        c.tile[lane] = gv0;
        c.tile[64 + lane] = gv1;
        wave_lds_fence();
        divdp[r] = (d_da_slot<COEF_LDS>(c, 0) + d_db_slot<COEF_LDS>(c, 1)) * rmetdet * rrearth;  // S:81-85
      } else if (MFMA) {
        divdp[r] = divergence_sphere_mfma(mc, Dinv, metdet, rmetdet, rrearth, x.uv.x * x.dp, x.uv.y * x.dp);  // P:114-121
      } else {
        divdp[r] = divergence_sphere<COEF_LDS>(c, lane, Dinv, metdet, rmetdet, rrearth, x.uv.x * x.dp, x.uv.y * x.dp);  // P:114-121
      }
      if (!RELOAD_T) T[r] = x.T;
      if (VADV && r == 0) s_Thalo[(w * 2 + 0) * PP + sl] = x.T;
      if (VADV && r == TPW - 1) s_Thalo[(w * 2 + 1) * PP + sl] = x.T;
      Tv[r] = MOIST ? x.T * (1.0 + k.rv_over_rd_m1 * (x.q * recip(x.dp))) : x.T;  // P:135,150-151
      run_dp += x.dp;
      run_div += divdp[r];
      if constexpr (!CARRY_IN) {  // (carried: they are there already)
        park_dp[r * PP] = x.dp;
        park_dp[BLK + r * PP] = x.uv.x;
        park_dp[2 * BLK + r * PP] = x.uv.y;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    s_tot_dp[w * PP + sl] = run_dp;
    s_tot_div[w * PP + sl] = run_div;
  }
  wg_sync();

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

  const double eta_zero = k.eta_ave_w * 0.0;  // eta_dot_dpdn_tmp == 0 (P:22,172)
  // A later call of a steady step loop: the previous call already added eta_ave_w * 0 to this element's eta_dot_dpdn, and
  // x + 0 is a fixed point after one application (it only turns -0 into +0) — unless eta_ave_w * 0 is a NaN, and then
  // the host does not take the step loop (caar_abi.hip try_fused_steps).  The read-modify-write that changes nothing is left out.
  constexpr bool eta_rmw = VADV || !CARRY_IN;
  struct LevelIn {
    dbl2 vnm1, vn0;
    double Tnm1, dpnm1, om, pec, eta, Tn0;
  };
  auto load_level = [&](int r) {
    const unsigned off = r * PP + ulane;
    LevelIn x;
    if constexpr (CARRY_M) {  // in cy->m*
      x.vnm1 = dbl2{0.0, 0.0};
      x.Tnm1 = x.dpnm1 = 0.0;
    } else {
      x.vnm1 = stream_load<SNT>(v_nm1 + off);
      x.Tnm1 = stream_load<SNT>(T_nm1 + off);
      x.dpnm1 = stream_load<SNT>(dp_nm1 + off);
    }
    x.vn0 = stream_load<SNT>(vn0 + off);
    x.om = stream_load<SNT>(omega_p + off);
    x.pec = stream_load<SNT>(pecnd + off);
    x.eta = eta_rmw ? stream_load<SNT>(eta + off) : 0.0;
    x.Tn0 = RELOAD_T ? stream_load<SNT>(T_n0 + off) : 0.0;
    return x;
  };
  static_assert(LA >= 1 && LA <= TPW, "look-ahead");
  LevelIn ahead[LA];  // requested before phase 2: in flight across it and the barrier
#pragma unroll
  for (int r = 0; r < LA; ++r) ahead[r] = load_level(r);

  // ---- phase 2: hydrostatic increments, their suffix sums inside the wave ---------------
  double base_dp = 0.0, base_div = 0.0;  // sums over the levels above this wave's first level
  for (int w2 = 0; w2 < w; ++w2) {
    base_dp += s_tot_dp[w2 * PP + sl];
    base_div += s_tot_div[w2 * PP + sl];
  }
  double sdot_sum = base_div;  // VADV: column total of divdp (X:237)
  if (VADV)
    for (int w2 = w; w2 < WAVES; ++w2) sdot_sum += s_tot_div[w2 * PP + sl];
  double wave_ht;  // sum of the hydrostatic increments over this wave's levels
  {
    double run = base_dp, acc = 0.0;
#pragma unroll
    for (int r = 0; r < TPW; ++r) {
      const double dpr = park_rd[r * PP];
      const double p = (k.p_top + run) + 0.5 * dpr;         // P:84,94-96 in closed form
      run += dpr;
      acc += (k.Rgas * Tv[r]) * (dpr * recip(p));           // Rgas*T_v*hkl, hkl = dp/p (P:300-302)
      __builtin_amdgcn_sched_barrier(0);
    }
    wave_ht = acc;
    s_tot_ht[w * PP + sl] = acc;
  }

  double l_eta_last = 0.0;
  if (tid < PP && eta_rmw) l_eta_last = eta_last[tid];
  wg_sync();

  // ---- phase 3: level-local tendencies and update, top level of the wave first ----------
  double below = 0.0;  // hydrostatic sum over the waves below this one, bottom-up (P:293,302)
  for (int w2 = WAVES - 1; w2 > w; --w2) below += s_tot_ht[w2 * PP + sl];

  double run_dp = base_dp, suml = base_div, run_ht = 0.0;
#pragma unroll
  for (int r = 0; r < TPW; ++r) {
    const unsigned off = r * PP + ulane;
    LevelIn cur = ahead[r % LA];
    if (r + LA < TPW) ahead[r % LA] = load_level(r + LA);
    if constexpr (CARRY_M) {
      cur.vnm1 = dbl2{cy->mu[r], cy->mv[r]};
      cur.Tnm1 = cy->mT[r];
      cur.dpnm1 = cy->mdp[r];
    }

    // The metric terms are re-read from LDS at every level instead of living in 26 registers
    // for the whole phase; the pointer is made opaque (but stays an LDS pointer: ds_read, not flat_load, which would
    // queue behind the outstanding global loads) so the loads are not hoisted.
    const lds_cptr geo = lds_reread_ptr(s_geo);
    const M22 Dinv = load_m22(geo + G_DINV, sl);
    const double phis = geo[G_PHIS + sl];
    const double dpr = park_rd[r * PP], ur = park_rd[BLK + r * PP], vr = park_rd[2 * BLK + r * PP];
    const double Tr = RELOAD_T ? cur.Tn0 : T[RELOAD_T ? 0 : r];

    const double p = (k.p_top + run_dp) + 0.5 * dpr;
    run_dp += dpr;
    const double rp = recip(p);
    const double ht = (k.Rgas * Tv[r]) * (dpr * rp);
    run_ht += ht;  // same increments, same order as in phase 2
    // levels below r inside this wave = wave total - inclusive prefix (P:302's phii)
    const double phi = (phis + (below + (wave_ht - run_ht))) + 0.5 * ht;  // P:303,309

    const M22 Dm = load_m22(geo + G_D, sl);
    const double Ephi = 0.5 * dot2(ur, ur, vr, vr) + phi + cur.pec;    // P:196
    double gp0, gp1, gT0, gT1, gE0, gE1, vort;
    if (BATCH) {
      // all five fields of the level go to their LDS slots, then every contraction reads:
      // one LDS round trip per level instead of one per operator
      const double vc0 = Dm.m00 * ur + Dm.m10 * vr;                    // S:106-107
      const double vc1 = Dm.m01 * ur + Dm.m11 * vr;                    // S:108-109
      wave_lds_fence();
      c.tile[lane] = p;
      c.tile[64 + lane] = Tr;
      c.tile[128 + lane] = Ephi;
      c.tile[192 + lane] = vc1;
      c.tile[256 + lane] = vc0;
      wave_lds_fence();
      const double pa = d_da_slot<COEF_LDS>(c, 0) * rrearth, pb = d_db_slot<COEF_LDS>(c, 0) * rrearth;  // S:34-35
      gp0 = Dinv.m00 * pa + Dinv.m10 * pb;                             // S:43-47
      gp1 = Dinv.m01 * pa + Dinv.m11 * pb;
      const double ta = d_da_slot<COEF_LDS>(c, 1) * rrearth, tb = d_db_slot<COEF_LDS>(c, 1) * rrearth;
      gT0 = Dinv.m00 * ta + Dinv.m10 * tb;
      gT1 = Dinv.m01 * ta + Dinv.m11 * tb;
      const double ea = d_da_slot<COEF_LDS>(c, 2) * rrearth, eb = d_db_slot<COEF_LDS>(c, 2) * rrearth;
      gE0 = Dinv.m00 * ea + Dinv.m10 * eb;
      gE1 = Dinv.m01 * ea + Dinv.m11 * eb;
      vort = (d_da_slot<COEF_LDS>(c, 3) - d_db_slot<COEF_LDS>(c, 4)) * rmetdet * rrearth;  // S:121-125
    } else if (MFMA) {
      gradient_sphere_mfma(mc, Dinv, rrearth, p, gp0, gp1);              // P:103
      vort = vorticity_sphere_mfma(mc, Dm, rmetdet, rrearth, ur, vr);    // P:122
      gradient_sphere_mfma(mc, Dinv, rrearth, Tr, gT0, gT1);             // P:200
      gradient_sphere_mfma(mc, Dinv, rrearth, Ephi, gE0, gE1);           // P:213
    } else {
      gradient_sphere<COEF_LDS>(c, lane, Dinv, rrearth, p, gp0, gp1);    // P:103
      vort = vorticity_sphere<COEF_LDS>(c, lane, Dm, rmetdet, rrearth, ur, vr);  // P:122
      gradient_sphere<COEF_LDS>(c, lane, Dinv, rrearth, Tr, gT0, gT1);   // P:200
      gradient_sphere<COEF_LDS>(c, lane, Dinv, rrearth, Ephi, gE0, gE1); // P:213
    }
    const double vgrad_p = dot2(ur, gp0, vr, gp1);                     // P:111
    const double ckk = 0.5 * rp, ckl = rp;                             // P:333-334
    const double om = __builtin_fma(-ckk, divdp[r], __builtin_fma(vgrad_p, rp, -(ckl * suml)));  // P:325,336,348
    double eta_lo = 0.0, eta_hi = 0.0, T_vadv = 0.0, u_vadv = 0.0, v_vadv = 0.0;
    if (VADV) {
      const int lev = lev0 + r;  // wave-uniform: hybi comes through scalar loads
      // X:238-254: eta_dot(k+1) = hybi(k+1)*sdot_sum - sum_{l<=k} divdp(l); 0 at the top and the surface
      const double e_lo = k.hybi[lev] * sdot_sum - suml;
      const double e_hi = k.hybi[lev + 1] * sdot_sum - (suml + divdp[r]);
      eta_lo = lev > 0 ? e_lo : 0.0;
      eta_hi = lev < NLEV - 1 ? e_hi : 0.0;
      const double half_rdp = 0.5 * recip(dpr);                        // X:118
      const double facp = half_rdp * eta_hi, facm = half_rdp * eta_lo; // CaarFunctor.hpp:526-527
      // Neighbouring levels, branch-free: at the top (bottom) level facm (facp) is exactly 0 and
      // the "neighbour" is some other finite value of the park (its last row is a zero pad).
      const int wu = w > 0 ? w - 1 : 0, wd = w < WAVES - 1 ? w + 1 : w;
      const double T_up = r > 0 ? T[r > 0 ? r - 1 : 0] : s_Thalo[(wu * 2 + 1) * PP + sl];
      const double T_dn = r < TPW - 1 ? T[r < TPW - 1 ? r + 1 : r] : s_Thalo[(wd * 2 + 0) * PP + sl];
      const double u_up = park_rd[BLK + (r - 1) * PP], u_dn = park_rd[BLK + (r + 1) * PP];
      const double v_up = park_rd[2 * BLK + (r - 1) * PP], v_dn = park_rd[2 * BLK + (r + 1) * PP];
      // CaarFunctor.hpp:513-546
      T_vadv = facp * (T_dn - Tr) + facm * (Tr - T_up);
      u_vadv = facp * (u_dn - ur) + facm * (ur - u_up);
      v_vadv = facp * (v_dn - vr) + facm * (vr - v_up);
    }
    suml += divdp[r];                                                  // P:339
    const double vgrad_T = dot2(ur, gT0, vr, gT1);                     // P:209
    const double gpterm = Tv[r] * rp;                                  // P:219
    const double glnps1 = k.Rgas * gpterm * gp0;                       // P:221
    const double glnps2 = k.Rgas * gpterm * gp1;                       // P:222
    const double fcor = geo[G_FCOR + sl], spheremp = geo[G_SPHEREMP + sl];
    double vtens1 = vr * (fcor + vort) - gE0 - glnps1;               // P:227
    double vtens2 = -ur * (fcor + vort) - gE1 - glnps2;              // P:228
    double ttens = -vgrad_T + k.kappa * Tv[r] * om;                    // P:230
    if (VADV) {
      vtens1 = -u_vadv + vr * (fcor + vort) - gE0 - glnps1;            // X:326-328
      vtens2 = -v_vadv - ur * (fcor + vort) - gE1 - glnps2;            // X:332-334
      ttens = -T_vadv - vgrad_T + k.kappa * Tv[r] * om;                // X:338
    }

    dbl2 vo;
    vo.x = spheremp * (cur.vnm1.x + k.dt2 * vtens1);                   // P:251
    vo.y = spheremp * (cur.vnm1.y + k.dt2 * vtens2);                   // P:252
    if (!STEPS || store_state) stream_store<SNT>(v_np1 + off, vo);
    const double T_new = spheremp * (cur.Tnm1 + k.dt2 * ttens);       // P:253
    if (!STEPS || store_state) stream_store<SNT>(T_np1 + off, T_new);
    const double dp_new = VADV ? spheremp * (cur.dpnm1 - k.dt2 * (divdp[r] + eta_hi - eta_lo))  // X:515-517
                               : spheremp * (cur.dpnm1 - k.dt2 * divdp[r]);                     // P:254
    debug_check_dp3d(dp_new);  // CaarFunctor.hpp:82-97 check_dp3d (-DCAAR_DEBUG builds only)
    if (!STEPS || store_state) stream_store<SNT>(dp_np1 + off, dp_new);
    if (!STEPS || store_phi) stream_store<SNT>(phi_out + off, phi);
    stream_store<SNT>(omega_p + off, cur.om + k.eta_ave_w * om);                 // P:173
    dbl2 vn;
    vn.x = cur.vn0.x + k.eta_ave_w * (ur * dpr);                       // P:117
    vn.y = cur.vn0.y + k.eta_ave_w * (vr * dpr);                       // P:118
    stream_store<SNT>(vn0 + off, vn);
    if (eta_rmw) stream_store<SNT>(eta + off, cur.eta + (VADV ? k.eta_ave_w * eta_lo : eta_zero));  // P:172, X:271-272
    if constexpr (STEPS) {  // the state just stored at np1 is the next call's n0: it replaces this level's n0 state,
      cy->mdp[r] = dpr;     // which is the next call's nm1
      cy->mu[r] = ur;
      cy->mv[r] = vr;
      cy->mT[r] = Tr;
      park_dp[r * PP] = dp_new;
      park_dp[BLK + r * PP] = vo.x;
      park_dp[2 * BLK + r * PP] = vo.y;
      T[r] = T_new;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (tid < PP && eta_rmw) eta_last[tid] = l_eta_last + eta_zero;                 // P:181
}

template <int NLEV, int TPW, int MINW, bool MOIST, bool SNT, bool COEF_LDS, bool RELOAD_T, bool BATCH, bool VADV = false, bool MFMA = false, int LA = 1>
__global__ __launch_bounds__(NLEV / TPW * 64, MINW) void caar_np8_kernel(const KernelArgs k) {
  request_kernel_args(k);
  __shared__ Np8Lds<NLEV, TPW, BATCH, VADV, MFMA> lds;
  caar_np8_element<NLEV, TPW, MINW, MOIST, SNT, COEF_LDS, RELOAD_T, BATCH, VADV, MFMA, LA>(k, lds);
}

// caar_run_steps / caar_launch_steps as ONE launch for NP=8 (the MFMA form; see caar_np4_steps.hip): every
// workgroup makes all nsteps calls for its element.  With rotating, distinct time levels the first call loads everything
// and every later call finds dp3d, u, v at n0 in the LDS park and T in registers, where the previous call left its np1
// results, and its nm1 state — the n0 state of the call before — in registers too (36 doubles per lane: a 512-thread
// workgroup alone on its CU may use 256 VGPRs); the state is therefore stored by the last three calls only (the earlier
// stores would be overwritten unread).  Default cache policy, so that the accumulators are re-read from the L2 that
// holds them.  Bit-identical to nsteps single launches.
template <int NLEV, int TPW, int MINW, bool MOIST, bool SNT, int LA>
__global__ __launch_bounds__(NLEV / TPW * 64, MINW) void caar_np8_steps_kernel(const KernelArgs k0, int nsteps, int rotate) {
  __shared__ Np8Lds<NLEV, TPW, false, false, true, true> lds;
  if (element_of_block(k0, blockIdx.x) < 0) return;
  int n0 = k0.n0, np1 = k0.np1, nm1 = k0.nm1;
  Np8Carry<TPW> cy;  // handed from call to call
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
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  if (steady) {
    caar_np8_element<NLEV, TPW, MINW, MOIST, SNT, false, false, false, false, true, LA, true, 0>(args(), lds, &cy, nsteps == 1, nsteps <= 3);
    for (int s = 1; s < nsteps; ++s) {
      rotate_levels();
      // (no barrier: the only LDS data a wave reads from other waves are the wave totals, and alternate calls use
      // alternate sets — see caar_np4_steps.hip)
      caar_np8_element<NLEV, TPW, MINW, MOIST, SNT, false, false, false, false, true, LA, true, 2>(args(), lds, &cy, s == nsteps - 1, s >= nsteps - 3, s & 1);
    }
  } else {
    for (int s = 0; s < nsteps; ++s) {
      caar_np8_element<NLEV, TPW, MINW, MOIST, SNT, false, false, false, false, true, LA, true, 0>(args(), lds, &cy, s == nsteps - 1);
      if (rotate) rotate_levels();
      lds_barrier();
    }
  }
}

template <int NLEV, int TPW, int MINW, bool SNT, int LA>
static hipError_t launch_np8_steps(const KernelArgs& k, int num_elems, int nsteps, int rotate, hipStream_t stream) {
  constexpr int THREADS = NLEV / TPW * 64;
  if (k.vadv) return hipErrorNotSupported;
  const int grid = k.per_xcd ? 8 * k.per_xcd : num_elems;
  if (k.qn0 >= 0) hipLaunchKernelGGL((caar_np8_steps_kernel<NLEV, TPW, MINW, true, SNT, LA>), dim3(grid), dim3(THREADS), 0, stream, k, nsteps, rotate);
  else hipLaunchKernelGGL((caar_np8_steps_kernel<NLEV, TPW, MINW, false, SNT, LA>), dim3(grid), dim3(THREADS), 0, stream, k, nsteps, rotate);
  return hipGetLastError();
}

template <int NLEV, int TPW, int MINW, bool NT, bool COEF_LDS = false, bool RELOAD_T = false, bool BATCH = false, bool MFMA = false, int LA = 1>
static hipError_t launch_np8(const KernelArgs& k, int num_elems, hipStream_t stream) {
  constexpr int THREADS = NLEV / TPW * 64;
  const int grid = k.per_xcd ? 8 * k.per_xcd : num_elems;
  if (k.vadv) {  // rsplit == 0: T stays in registers (no RELOAD_T form); every variant takes the MFMA contractions (the
                 // Eulerian form of the LDS-tile comparators spilled 6 VGPRs, and nothing compares Eulerian forms)
    if (k.qn0 >= 0)
      hipLaunchKernelGGL((caar_np8_kernel<NLEV, TPW, MINW, true, NT, false, false, false, true, true>), dim3(grid), dim3(THREADS), 0, stream, k);
    else
      hipLaunchKernelGGL((caar_np8_kernel<NLEV, TPW, MINW, false, NT, false, false, false, true, true>), dim3(grid), dim3(THREADS), 0, stream, k);
    return hipGetLastError();
  }
  if (k.qn0 >= 0)
    hipLaunchKernelGGL((caar_np8_kernel<NLEV, TPW, MINW, true, NT, COEF_LDS, RELOAD_T, BATCH, false, MFMA, LA>), dim3(grid), dim3(THREADS), 0, stream, k);
  else  // dry branch (P:128-139)
    hipLaunchKernelGGL((caar_np8_kernel<NLEV, TPW, MINW, false, NT, COEF_LDS, RELOAD_T, BATCH, false, MFMA, LA>), dim3(grid), dim3(THREADS), 0, stream, k);
  return hipGetLastError();
}

// (non-const on purpose, see caar_np4.hip)
KernelVariant kNp8Nlev72[] = {
    {"caar_np8_kernel<72, 9, 1, true, true, false, false, false, false, true, 2>", "8 waves x 9 levels, nt, MFMA contractions, update-phase inputs requested two levels ahead", launch_np8<72, 9, 1, true, false, false, false, true, 2>, true, launch_np8_steps<72, 9, 1, false, 2>},
    {"caar_np8_kernel<72, 9, 1, true, true, false, false, false, false, true, 1>", "8 waves x 9 levels, nt, Dvv contractions on v_mfma_f64_4x4x4 (lane = MFMA result layout, no LDS tile)", launch_np8<72, 9, 1, true, false, false, false, true>, false, launch_np8_steps<72, 9, 1, true, 1>},
    {"caar_np8_kernel<72, 9, 1, true, true, true, false, false, false, false, 1>", "8 waves x 9 levels, nt, Dvv slices re-read from LDS", launch_np8<72, 9, 1, true, true, false, false>},
    {"caar_np8_kernel<72, 9, 1, true, false, true, false, false, false, false, 1>", "8 waves x 9 levels, default cache policy", launch_np8<72, 9, 1, false, true, false, false>},
    {"caar_np8_kernel<72, 12, 1, true, true, false, false, false, false, true, 1>", "6 waves x 12 levels, nt, MFMA contractions", launch_np8<72, 12, 1, true, false, false, false, true>},
};
int kNp8Nlev72Count = sizeof(kNp8Nlev72) / sizeof(kNp8Nlev72[0]);

#ifdef CAAR_DEBUG
long long debug_dp3d_count_np8(int reset) { return debug_dp3d_count_of_this_tu(reset); }
#endif

}  // namespace caar
