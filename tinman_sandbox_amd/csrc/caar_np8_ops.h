// caar_np8_ops.h — the three sphere operators for NP=8: one level = one wavefront.  Two forms of the 8x8 Dvv
// contractions: the DEFAULT "MFMA form" (v_mfma_f64_4x4x4, lane = MFMA result layout, no LDS tile; further down) and
// the direct form through a wave-private 64-double LDS tile (the comparator variant; first in this file).
// Reference: cxx/pointers_only/sphere_operators.cpp:9-129 (S:).
#pragma once

#include <hip/hip_runtime.h>

#include "caar_kernel_args.h"

namespace caar {

namespace np8 {

constexpr int NP = 8, PP = 64;
enum { G_FCOR = 0, G_SPHEREMP = 64, G_METDET = 128, G_RMETDET = 192, G_PHIS = 256, G_D = 320, G_DINV = 576, G_SIZE = 832 };

// wave-private LDS is written and read by different lanes of the same wave: LDS
// instructions of one wave execute in order, the fence only pins the compiler.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Dvv slices of lane (a, b): ca[k] = Dvv[k][a], cb[k] = Dvv[k][b].  Either held in
// registers (32 VGPRs) or re-read from the transposed LDS copy dvvT[j*8+k] = Dvv[k][j] at
// every use (4 x ds_read_b128 per slice) when the kernel needs the registers.
struct Ctx {
  double ca[NP];
  double cb[NP];
  const double* dvvT;  // LDS copy, or nullptr when ca/cb are valid
  double* tile;        // LDS, this wave's 64-double tile
  int a, b;
};

// sum_k Dvv[k][a] f[k][b] and sum_k Dvv[k][b] f[a][k] of the field currently in c.tile
template <bool COEF_LDS>
__device__ __forceinline__ double d_da_tile(const Ctx& c) {
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < NP; ++k) s += (COEF_LDS ? c.dvvT[c.a * NP + k] : c.ca[k]) * c.tile[k * NP + c.b];
  return s;
}
template <bool COEF_LDS>
__device__ __forceinline__ double d_db_tile(const Ctx& c) {
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < NP; ++k) s += (COEF_LDS ? c.dvvT[c.b * NP + k] : c.cb[k]) * c.tile[c.a * NP + k];
  return s;
}
// The same two contractions of the field stored in tile slot j (c.tile + 64*j): the batched
// form writes all fields of a level into their own slots first, so the LDS round trip is
// paid once per level instead of once per operator.
template <bool COEF_LDS>
__device__ __forceinline__ double d_da_slot(const Ctx& c, int j) {
  const double* t = c.tile + 64 * j;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < NP; ++k) s += (COEF_LDS ? c.dvvT[c.a * NP + k] : c.ca[k]) * t[k * NP + c.b];
  return s;
}
template <bool COEF_LDS>
__device__ __forceinline__ double d_db_slot(const Ctx& c, int j) {
  const double* t = c.tile + 64 * j;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < NP; ++k) s += (COEF_LDS ? c.dvvT[c.b * NP + k] : c.cb[k]) * t[c.a * NP + k];
  return s;
}

__device__ __forceinline__ void put_tile(const Ctx& c, int lane, double f) {
  wave_lds_fence();  // earlier reads of the tile are done
  c.tile[lane] = f;
  wave_lds_fence();
}

struct M22 {
  double m00, m01, m10, m11;
};
// The 2x2 tensors lie in LDS component-major, [r*2 + c][point] (caar_np8.hip stages them that way): the 64 lanes of a
// wave read 64 consecutive doubles per component.  Point-major (32-byte stride between lanes) every one of these reads —
// eight per level and lane in the last phase — was an 8-way bank conflict: SQ_LDS_BANK_CONFLICT 10 900 cycles per
// element-call against 4 650 LDS-active ones (profiles/r03/pmc_issue.json).
// `pt`: this lane's slot inside a 64-entry table — the kernels stage every per-point table in the order their lanes hold
// the points (MFMA form: slot = lane, the table is permuted when it is staged), so the 64 lanes read 64 consecutive doubles.
// Indexed by the GLL point itself in the MFMA lane mapping (point = 32I + 8h + b for lane = 16h + 8I + b) the two 8-lane
// groups I = 0, 1 of a 16-lane row lie 32 doubles = 64 banks apart: every read was a 2-way conflict.
template <class Ptr>  // const double* or lds_cptr
__device__ __forceinline__ M22 load_m22(Ptr g, int pt) {
  M22 m;
  m.m00 = g[0 * PP + pt];
  m.m01 = g[1 * PP + pt];
  m.m10 = g[2 * PP + pt];
  m.m11 = g[3 * PP + pt];
  return m;
}

// gradient_sphere, S:9-48
template <bool COEF_LDS = false>
__device__ __forceinline__ void gradient_sphere(const Ctx& c, int lane, const M22& Dinv, double rrearth,
                                                double s, double& g0, double& g1) {
  put_tile(c, lane, s);
  const double v1 = d_da_tile<COEF_LDS>(c) * rrearth;
  const double v2 = d_db_tile<COEF_LDS>(c) * rrearth;
  g0 = dot2(Dinv.m00, v1, Dinv.m10, v2);
  g1 = dot2(Dinv.m01, v1, Dinv.m11, v2);
}
// divergence_sphere, S:50-89
template <bool COEF_LDS = false>
__device__ __forceinline__ double divergence_sphere(const Ctx& c, int lane, const M22& Dinv, double metdet,
                                                    double rmetdet, double rrearth, double u, double v) {
  const double gv0 = metdet * dot2(Dinv.m00, u, Dinv.m01, v);
  const double gv1 = metdet * dot2(Dinv.m10, u, Dinv.m11, v);
  put_tile(c, lane, gv0);
  const double dudx = d_da_tile<COEF_LDS>(c);
  put_tile(c, lane, gv1);
  const double dvdy = d_db_tile<COEF_LDS>(c);
  return (dudx + dvdy) * rmetdet * rrearth;
}
// vorticity_sphere, S:91-129
template <bool COEF_LDS = false>
__device__ __forceinline__ double vorticity_sphere(const Ctx& c, int lane, const M22& D, double rmetdet,
                                                   double rrearth, double u, double v) {
  const double vc0 = dot2(D.m00, u, D.m10, v);
  const double vc1 = dot2(D.m01, u, D.m11, v);
  put_tile(c, lane, vc1);
  const double dvdx = d_da_tile<COEF_LDS>(c);
  put_tile(c, lane, vc0);
  const double dudy = d_db_tile<COEF_LDS>(c);
  return (dvdx - dudy) * rmetdet * rrearth;
}

// ---------------------------------------------------------------------------------------------------
// MFMA form of the two 8x8 contractions (BASELINE north_star: "MFMA used only for the batched Dvv
// contraction when NP >= 8"): v_mfma_f64_4x4x4_4b_f64, four independent 4x4x4 products per issue, so an
// 8x8 * 8x8 product is its 2x2 output blocks x 2 k-blocks = two issues, no padding, no wasted lanes
// (the 16x16x4 form would carry an 8-row operand in a 16-row tile: half of every issue multiplies zeros).
// Operand layouts measured on gfx950 (tools/probes/mfma_f64_probe.hip, profiles/r02/mfma_probe.log):
//     A[blk][i][k] in lane 16k + 4blk + i,   B[blk][k][j] in lane 16k + 4blk + j,   D[blk][i][j] in lane 16i + 4blk + j.
// With blk = 2I + J the result D_(I,J)[i][j] = M[4I+i][4J+j] of an 8x8 matrix M sits in lane 16i + 8I + 4J + j.
// The kernel adopts exactly that as its lane -> GLL point mapping (mfma_point): results land where the
// pointwise code needs them, every 8-lane group is still one contiguous 64-byte row of the level.
//   d/da = Dvv^T . F :  A = blocks of Dvv^T (two per-lane constants), B = blocks of F: lane (k, I, J, j) needs
//          F[4K+k][4J+j], i.e. its own value or the one 8 lanes away in its 16-lane row -> one bank-masked
//          DPP row_ror:8 per k-block, no LDS.
//   d/db = F . Dvv   :  B = blocks of Dvv (two per-lane constants), A = blocks of F in the TRANSPOSED in-block
//          placement (row index in the low lane bits): one ds_bpermute per k-block (crossbar only, no LDS memory).
// Against the direct form per field pair: 4 MFMA + 4 DPP movs + 4 ds_bpermute_b32 instead of 16 fp64 FMAs,
// one ds_write_b64, four ds_read_b128 and eight ds_read_b64.
__device__ __forceinline__ int mfma_point(int lane) {
  const int a = 4 * ((lane >> 3) & 1) + (lane >> 4), b = lane & 7;
  return a * NP + b;
}
// ... and its inverse: the lane that holds GLL point p = a*8 + b
__device__ __forceinline__ int mfma_lane_of_point(int p) {
  const int a = p >> 3, b = p & 7;
  return 16 * (a & 3) + 8 * (a >> 2) + b;
}

struct MfmaCtx {
  double a_da[2];  // A operand of d/da for k-block K: Dvv[4K + h][4I + l]     (lane = 16h + 8I + 4J + l)
  double b_db[2];  // B operand of d/db for k-block K: Dvv[4K + h][4J + l]
  int src_db[2];   // lane that holds F[4I + l][4K + h]: 16l + 8I + 4K + h
};

__device__ __forceinline__ MfmaCtx make_mfma_ctx(const double* dvv /* Dvv[k][j] row-major, any address space */, int lane) {
  const int h = lane >> 4, I = (lane >> 3) & 1, J = (lane >> 2) & 1, l = lane & 3;
  MfmaCtx c;
#pragma unroll
  for (int K = 0; K < 2; ++K) {
    c.a_da[K] = dvv[(4 * K + h) * NP + 4 * I + l];
    c.b_db[K] = dvv[(4 * K + h) * NP + 4 * J + l];
    c.src_db[K] = 16 * l + 8 * I + 4 * K + h;
  }
  return c;
}

__device__ __forceinline__ double mfma4(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// The two B operands of d/da come from ONE unmasked row_ror:8 (the value 8 lanes away in the row, for every lane: two
// v_mov_b32_dpp into a fresh register) and two selects by the lane's half of the row (v_cndmask with a constant lane mask):
// 6 VALU instructions, pure data movement.  (Rounds 2-3 used two bank-masked DPP moves, 8: docs/EXPERIMENTS.md A.)
// sum_k Dvv[k][a] f[k][b] at this lane's point (a, b)
__device__ __forceinline__ double mfma_d_da(const MfmaCtx& c, double f) {
  const double p = __builtin_amdgcn_update_dpp(0.0, f, 0x128, 0xf, 0xf, true);  // row_ror:8, every lane has a source
  const bool upper = (__lane_id() & 8) != 0;                                    // I = 1: lanes 8..15 of each row
  const double b0 = upper ? p : f;  // block row K = 0: lanes 8..15 take lane - 8's value
  const double b1 = upper ? f : p;  // block row K = 1: lanes 0..7 take lane + 8's value
  return mfma4(c.a_da[1], b1, mfma4(c.a_da[0], b0, 0.0));
}
// sum_k Dvv[k][b] f[a][k]
__device__ __forceinline__ double mfma_d_db(const MfmaCtx& c, double f) {
  const double a0 = __shfl(f, c.src_db[0], 64);
  const double a1 = __shfl(f, c.src_db[1], 64);
  return mfma4(a1, c.b_db[1], mfma4(a0, c.b_db[0], 0.0));
}

// the three operators, S:9-129, on top of the MFMA contractions
__device__ __forceinline__ void gradient_sphere_mfma(const MfmaCtx& c, const M22& Dinv, double rrearth, double s,
                                                     double& g0, double& g1) {
  const double v1 = mfma_d_da(c, s) * rrearth;
  const double v2 = mfma_d_db(c, s) * rrearth;
  g0 = dot2(Dinv.m00, v1, Dinv.m10, v2);
  g1 = dot2(Dinv.m01, v1, Dinv.m11, v2);
}
__device__ __forceinline__ double divergence_sphere_mfma(const MfmaCtx& c, const M22& Dinv, double metdet, double rmetdet,
                                                         double rrearth, double u, double v) {
  const double gv0 = metdet * dot2(Dinv.m00, u, Dinv.m01, v);
  const double gv1 = metdet * dot2(Dinv.m10, u, Dinv.m11, v);
  return (mfma_d_da(c, gv0) + mfma_d_db(c, gv1)) * rmetdet * rrearth;
}
__device__ __forceinline__ double vorticity_sphere_mfma(const MfmaCtx& c, const M22& D, double rmetdet, double rrearth,
                                                        double u, double v) {
  const double vc0 = dot2(D.m00, u, D.m10, v);
  const double vc1 = dot2(D.m01, u, D.m11, v);
  return (mfma_d_da(c, vc1) - mfma_d_db(c, vc0)) * rmetdet * rrearth;
}

}  // namespace np8

}  // namespace caar
