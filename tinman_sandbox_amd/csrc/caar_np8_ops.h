// caar_np8_ops.h — the three sphere operators for NP=8: one level = one wavefront, the
// 8x8 Dvv contractions go through a wave-private 64-double LDS tile (see caar_np8.hip).
// Reference: cxx/pointers_only/sphere_operators.cpp:9-129 (S:).
#ifndef CAAR_NP8_OPS_H
#define CAAR_NP8_OPS_H

#include <hip/hip_runtime.h>

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
__device__ __forceinline__ M22 load_m22(const double* g, int pt) {
  M22 m;
  m.m00 = g[pt * 4 + 0];
  m.m01 = g[pt * 4 + 1];
  m.m10 = g[pt * 4 + 2];
  m.m11 = g[pt * 4 + 3];
  return m;
}

// gradient_sphere, S:9-48
template <bool COEF_LDS = false>
__device__ __forceinline__ void gradient_sphere(const Ctx& c, int lane, const M22& Dinv, double rrearth,
                                                double s, double& g0, double& g1) {
  put_tile(c, lane, s);
  const double v1 = d_da_tile<COEF_LDS>(c) * rrearth;
  const double v2 = d_db_tile<COEF_LDS>(c) * rrearth;
  g0 = Dinv.m00 * v1 + Dinv.m10 * v2;
  g1 = Dinv.m01 * v1 + Dinv.m11 * v2;
}
// divergence_sphere, S:50-89
template <bool COEF_LDS = false>
__device__ __forceinline__ double divergence_sphere(const Ctx& c, int lane, const M22& Dinv, double metdet,
                                                    double rmetdet, double rrearth, double u, double v) {
  const double gv0 = metdet * (Dinv.m00 * u + Dinv.m01 * v);
  const double gv1 = metdet * (Dinv.m10 * u + Dinv.m11 * v);
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
  const double vc0 = D.m00 * u + D.m10 * v;
  const double vc1 = D.m01 * u + D.m11 * v;
  put_tile(c, lane, vc1);
  const double dvdx = d_da_tile<COEF_LDS>(c);
  put_tile(c, lane, vc0);
  const double dudy = d_db_tile<COEF_LDS>(c);
  return (dvdx - dudy) * rmetdet * rrearth;
}

}  // namespace np8

}  // namespace caar
#endif
