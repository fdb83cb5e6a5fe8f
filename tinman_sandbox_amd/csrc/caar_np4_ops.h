// caar_np4_ops.h — the three sphere operators for NP=4, in two forms.
//
// gradient_sphere / divergence_sphere / vorticity_sphere of the reference
// (compute_and_apply_rhs_test/cxx/pointers_only/sphere_operators.cpp:9-129, cited as S:):
//   * DPP form: lane = a*4 + b of a 16-lane row holding one level of one element (the simple stand-alone operators,
//     caar_operators.hip / caar_operators_ex.hip);
//   * MFMA form (second half of this file): the four levels of a tile through one v_mfma_f64_4x4x4 per contraction —
//     the fused kernels (caar_np4_kernel.h, default) and the composite operators.
#pragma once

#include <hip/hip_runtime.h>

namespace caar {

// ---------------------------------------------------------------- DPP helpers
// dpp_ctrl encodings (LLVM AMDGPU): quad_perm = p0|p1<<2|p2<<4|p3<<6,
// row_ror:n = 0x120+n.  All lanes have a valid source for these controls.
template <int CTRL>
__device__ __forceinline__ double dpp(double x) {
  return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ int dppi(int x) {
  return __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, true);
}

// Per-lane slices of Dvv for lane (a, b) of a 16-lane row:
//   ca[r] = Dvv[k_r][a] where k_r is the `a` index of the lane that row_ror:(4r)
//           delivers to this lane (found by rotating the lane id itself, so the
//           code does not depend on the rotate direction),
//   cb[k] = Dvv[k][b].
struct RowCoef {
  double ca[4];
  double cb[4];
};

// sum_k Dvv[k][a] f[k][b]   (derivative along the first GLL index; S:30,81,121)
__device__ __forceinline__ double d_da(const RowCoef& c, double f) {
  double s = c.ca[0] * f;
  s += c.ca[1] * dpp<0x124>(f);
  s += c.ca[2] * dpp<0x128>(f);
  s += c.ca[3] * dpp<0x12C>(f);
  return s;
}
// sum_k Dvv[k][b] f[a][k]   (derivative along the second GLL index; S:31,82,122)
__device__ __forceinline__ double d_db(const RowCoef& c, double f) {
  double s = c.cb[0] * dpp<0x00>(f);
  s += c.cb[1] * dpp<0x55>(f);
  s += c.cb[2] * dpp<0xAA>(f);
  s += c.cb[3] * dpp<0xFF>(f);
  return s;
}

// Metric 2x2 of this lane's point, row-major m[r][c] -> {m00, m01, m10, m11}.
struct M22 {
  double m00, m01, m10, m11;
};

// gradient_sphere, S:9-48
__device__ __forceinline__ void gradient_sphere(const RowCoef& c, const M22& Dinv, double rrearth,
                                                double s, double& g0, double& g1) {
  const double v1 = d_da(c, s) * rrearth;
  const double v2 = d_db(c, s) * rrearth;
  g0 = dot2(Dinv.m00, v1, Dinv.m10, v2);
  g1 = dot2(Dinv.m01, v1, Dinv.m11, v2);
}
// divergence_sphere, S:50-89
__device__ __forceinline__ double divergence_sphere(const RowCoef& c, const M22& Dinv, double metdet,
                                                    double rmetdet, double rrearth, double u, double v) {
  const double gv0 = metdet * dot2(Dinv.m00, u, Dinv.m01, v);
  const double gv1 = metdet * dot2(Dinv.m10, u, Dinv.m11, v);
  return (d_da(c, gv0) + d_db(c, gv1)) * rmetdet * rrearth;
}
// vorticity_sphere, S:91-129
__device__ __forceinline__ double vorticity_sphere(const RowCoef& c, const M22& D, double rmetdet,
                                                   double rrearth, double u, double v) {
  const double vc0 = dot2(D.m00, u, D.m10, v);
  const double vc1 = dot2(D.m01, u, D.m11, v);
  return (d_da(c, vc1) - d_db(c, vc0)) * rmetdet * rrearth;
}

// Per-lane Dvv slices for lane `lane` from a 16-entry Dvv table (LDS or global).
__device__ __forceinline__ RowCoef make_row_coef(const double* dvv, int lane) {
  RowCoef c;
  const int pt = lane & 15, a = pt >> 2, b = pt & 3;
  const int s1 = dppi<0x124>(lane), s2 = dppi<0x128>(lane), s3 = dppi<0x12C>(lane);
  c.ca[0] = dvv[a * 4 + a];
  c.ca[1] = dvv[((s1 >> 2) & 3) * 4 + a];
  c.ca[2] = dvv[((s2 >> 2) & 3) * 4 + a];
  c.ca[3] = dvv[((s3 >> 2) & 3) * 4 + a];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) c.cb[kk] = dvv[kk * 4 + b];
  return c;
}

// ---------------------------------------------------------------------------------------------------
// MFMA form of the NP=4 contractions: what the fused kernels use (caar_np4_kernel.h) and what
// the composite operators of caar_operators_ex.hip need (2-6 contraction pairs per point for 1-2 KiB of traffic per tile:
// bound by VALU issue in the DPP form, which costs 14 cross-lane moves + 8 FMAs per pair).  v_mfma_f64_4x4x4_4b_f64 multiplies four independent 4x4 blocks per
// issue: one block = one LEVEL of the tile, so one issue is one contraction of all four levels a wave holds.
// Operand layouts (measured: tools/probes/mfma_f64_probe.hip, profiles/r02/mfma_probe.log):
//     A[blk][i][k] in lane 16k + 4blk + i,   B[blk][k][j] in lane 16k + 4blk + j,   D[blk][i][j] in lane 16i + 4blk + j.
// A kernel that uses these adopts lane = 16a + 4lev + b as its lane -> (GLL point, level) mapping (mfma4_point /
// mfma4_level; the wave still covers the same 64 consecutive doubles of the layout [lev][a][b]): a field F in that
// mapping IS a B operand and a result, so
//   da(F) = Dvv^T . F : A = Dvv^T (a per-lane constant), B = F              -> one MFMA, no cross-lane move at all
//   db(F) = F . Dvv   : A = F in the transposed in-block placement (one 64-bit ds_bpermute), B = Dvv (constant)
//   wa, wb            : the same with the constants taken from Dvv^T.
__device__ __forceinline__ int mfma4_point(int lane) { return (lane >> 4) * 4 + (lane & 3); }  // a*4 + b
__device__ __forceinline__ int mfma4_level(int lane) { return (lane >> 2) & 3; }               // level inside the tile
struct Mfma4Ctx {
  double d_hl;  // Dvv[h][l] for lane = 16h + 4blk + l: A operand of da, B operand of db
  double d_lh;  // Dvv[l][h]: A operand of wa, B operand of wb
  int src_t;    // the lane that holds F[l][h] of this lane's level: 16l + 4blk + h
};
__device__ __forceinline__ Mfma4Ctx make_mfma4_ctx(const double* dvv /* Dvv[k][j] row-major */, int lane) {
  const int h = lane >> 4, blk = (lane >> 2) & 3, l = lane & 3;
  Mfma4Ctx c;
  c.d_hl = dvv[h * 4 + l];
  c.d_lh = dvv[l * 4 + h];
  c.src_t = 16 * l + 4 * blk + h;
  return c;
}
__device__ __forceinline__ double mfma4x4(double a, double b) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0); }
// sum_k Dvv[k][a] f[k][b]
__device__ __forceinline__ double mfma4_d_da(const Mfma4Ctx& c, double f) { return mfma4x4(c.d_hl, f); }
// sum_k Dvv[k][b] f[a][k]
// (one 64-bit ds_bpermute moves F into the transposed in-block placement; the DPP / bank-spread alternatives: docs/EXPERIMENTS.md A)
__device__ __forceinline__ double mfma4_d_db(const Mfma4Ctx& c, double f) { return mfma4x4(__shfl(f, c.src_t, 64), c.d_hl); }
// sum_k Dvv[a][k] f[k][b]
__device__ __forceinline__ double mfma4_w_a(const Mfma4Ctx& c, double f) { return mfma4x4(c.d_lh, f); }
// sum_k Dvv[b][k] f[a][k]
__device__ __forceinline__ double mfma4_w_b(const Mfma4Ctx& c, double f) { return mfma4x4(__shfl(f, c.src_t, 64), c.d_lh); }


// the three operators of the path (S:9-129) on the MFMA contractions: the same formulas as the DPP forms above
__device__ __forceinline__ void gradient_sphere(const Mfma4Ctx& c, const M22& Dinv, double rrearth, double s, double& g0, double& g1) {
  const double v1 = mfma4_d_da(c, s) * rrearth;
  const double v2 = mfma4_d_db(c, s) * rrearth;
  g0 = dot2(Dinv.m00, v1, Dinv.m10, v2);
  g1 = dot2(Dinv.m01, v1, Dinv.m11, v2);
}
__device__ __forceinline__ double divergence_sphere(const Mfma4Ctx& c, const M22& Dinv, double metdet, double rmetdet, double rrearth,
                                                    double u, double v) {
  const double gv0 = metdet * dot2(Dinv.m00, u, Dinv.m01, v);
  const double gv1 = metdet * dot2(Dinv.m10, u, Dinv.m11, v);
  return (mfma4_d_da(c, gv0) + mfma4_d_db(c, gv1)) * rmetdet * rrearth;
}
__device__ __forceinline__ double vorticity_sphere(const Mfma4Ctx& c, const M22& D, double rmetdet, double rrearth, double u, double v) {
  const double vc0 = dot2(D.m00, u, D.m10, v);
  const double vc1 = dot2(D.m01, u, D.m11, v);
  return (mfma4_d_da(c, vc1) - mfma4_d_db(c, vc0)) * rmetdet * rrearth;
}

}  // namespace caar
