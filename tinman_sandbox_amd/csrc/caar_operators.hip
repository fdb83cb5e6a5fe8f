// caar_operators.hip — the three sphere operators as stand-alone device entry points.
//
// The reference exposes gradient_sphere / divergence_sphere / vorticity_sphere as functions
// (cxx/pointers_only/sphere_operators.hpp:9-16: one np x np field of one element in, one
// out).  In the fused CAAR kernels they are device functions without a launch of their
// own; this file launches the SAME device functions (caar_np4_ops.h, caar_np8_ops.h) on a
// batch of levels, so the operators can be driven and parity-tested exactly like the
// reference's (caar_sphere_operator in include/caar.h).
#include <hip/hip_runtime.h>

#include "caar_kernel_args.h"
#include "caar_np4_ops.h"
#include "caar_np8_ops.h"

namespace caar {

// which: 0 gradient (in [lev][np][np] -> out [lev][np][np][2]), 1 divergence, 2 vorticity
// (in [lev][np][np][2] -> out [lev][np][np]).  One wave per 64 points.
__global__ void sphere_operator_np4(int which, const double* __restrict__ in, double* __restrict__ out,
                                    const double* __restrict__ D, const double* __restrict__ Dinv,
                                    const double* __restrict__ metdet, const double* __restrict__ rmetdet,
                                    const double* __restrict__ dvv, int ie, int nlevels, double rrearth) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int pt = lane & 15;
  const int lev = wave * 4 + (lane >> 4);
  const bool live = lev < nlevels;
  const RowCoef c = make_row_coef(dvv, lane);
  const size_t g = (size_t)ie * 16 + pt;
  M22 Di, Dm;
  Di.m00 = Dinv[g * 4 + 0]; Di.m01 = Dinv[g * 4 + 1]; Di.m10 = Dinv[g * 4 + 2]; Di.m11 = Dinv[g * 4 + 3];
  Dm.m00 = D[g * 4 + 0]; Dm.m01 = D[g * 4 + 1]; Dm.m10 = D[g * 4 + 2]; Dm.m11 = D[g * 4 + 3];
  const size_t o = (size_t)lev * 16 + pt;
  // every lane runs the DPP code (rows of a dead level just compute on zeros)
  if (which == 0) {
    const double s = live ? in[o] : 0.0;
    double g0, g1;
    gradient_sphere(c, Di, rrearth, s, g0, g1);
    if (live) { out[2 * o] = g0; out[2 * o + 1] = g1; }
  } else {
    const double u = live ? in[2 * o] : 0.0, v = live ? in[2 * o + 1] : 0.0;
    const double r = which == 1 ? divergence_sphere(c, Di, metdet[g], rmetdet[g], rrearth, u, v)
                                : vorticity_sphere(c, Dm, rmetdet[g], rrearth, u, v);
    if (live) out[o] = r;
  }
}

__global__ void sphere_operator_np8(int which, const double* __restrict__ in, double* __restrict__ out,
                                    const double* __restrict__ D, const double* __restrict__ Dinv,
                                    const double* __restrict__ metdet, const double* __restrict__ rmetdet,
                                    const double* __restrict__ dvv, int ie, int nlevels, double rrearth) {
  constexpr int NP = np8::NP;
  __shared__ __attribute__((aligned(16))) double s_tile[4 * 64];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int lev = blockIdx.x * (blockDim.x >> 6) + w;
  if (lev >= nlevels) return;  // wave-uniform
  np8::Ctx c;
  c.dvvT = nullptr;
  c.tile = s_tile + w * 64;
  c.a = lane >> 3;
  c.b = lane & 7;
#pragma unroll
  for (int kk = 0; kk < NP; ++kk) {
    c.ca[kk] = dvv[kk * NP + c.a];
    c.cb[kk] = dvv[kk * NP + c.b];
  }
  const size_t g = (size_t)ie * 64 + lane;
  const np8::M22 Di = np8::load_m22(Dinv + (size_t)ie * 256, lane), Dm = np8::load_m22(D + (size_t)ie * 256, lane);
  const size_t o = (size_t)lev * 64 + lane;
  if (which == 0) {
    double g0, g1;
    np8::gradient_sphere(c, lane, Di, rrearth, in[o], g0, g1);
    out[2 * o] = g0;
    out[2 * o + 1] = g1;
  } else if (which == 1) {
    out[o] = np8::divergence_sphere(c, lane, Di, metdet[g], rmetdet[g], rrearth, in[2 * o], in[2 * o + 1]);
  } else {
    out[o] = np8::vorticity_sphere(c, lane, Dm, rmetdet[g], rrearth, in[2 * o], in[2 * o + 1]);
  }
}

// recip() of caar_kernel_args.h on its own (numerics test hook: caar_reciprocal)
__global__ void reciprocal_kernel(const double* __restrict__ in, double* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = recip(in[i]);
}
hipError_t launch_reciprocal(const double* in, double* out, size_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const size_t want = (n + 255) / 256;
  hipLaunchKernelGGL(reciprocal_kernel, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, s, in, out, n);
  return hipGetLastError();
}

hipError_t launch_sphere_operator(int np, int which, const double* in, double* out, const double* D,
                                  const double* Dinv, const double* metdet, const double* rmetdet,
                                  const double* dvv, int ie, int nlevels, double rrearth, hipStream_t s) {
  if (nlevels <= 0) return hipSuccess;
  if (np == 4) {
    const int waves = (nlevels + 3) / 4, blocks = (waves + 3) / 4;
    hipLaunchKernelGGL(sphere_operator_np4, dim3(blocks), dim3(256), 0, s, which, in, out, D, Dinv, metdet, rmetdet,
                       dvv, ie, nlevels, rrearth);
  } else if (np == 8) {
    hipLaunchKernelGGL(sphere_operator_np8, dim3((nlevels + 3) / 4), dim3(256), 0, s, which, in, out, D, Dinv, metdet,
                       rmetdet, dvv, ie, nlevels, rrearth);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace caar
