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

// which: 0 gradient (in [e][lev][np][np] -> out [e][lev][np][np][2]), 1 divergence, 2 vorticity
// (in [e][lev][np][np][2] -> out [e][lev][np][np]), e = 0 .. ne-1 for elements ie0 .. ie0+ne-1.
// One workgroup per element (grid-stride), its waves walk the element's 64-point tiles, so the
// Dvv slices and the metric terms are fetched once per wave and element, not once per tile.
__global__ __launch_bounds__(256) void sphere_operator_np4(int which, const double* __restrict__ in, double* __restrict__ out,
                                    const double* __restrict__ D, const double* __restrict__ Dinv,
                                    const double* __restrict__ metdet, const double* __restrict__ rmetdet,
                                    const double* __restrict__ dvv, int ie0, int ne, int nlevels, double rrearth) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int pt = lane & 15, sub = lane >> 4;
  const RowCoef c = make_row_coef(dvv, lane);
  const int ntiles = (nlevels + 3) / 4;
  for (int e = blockIdx.x; e < ne; e += gridDim.x) {
    const size_t g = (size_t)(ie0 + e) * 16 + pt;
    M22 Di, Dm;
    Di.m00 = Dinv[g * 4 + 0]; Di.m01 = Dinv[g * 4 + 1]; Di.m10 = Dinv[g * 4 + 2]; Di.m11 = Dinv[g * 4 + 3];
    Dm.m00 = D[g * 4 + 0]; Dm.m01 = D[g * 4 + 1]; Dm.m10 = D[g * 4 + 2]; Dm.m11 = D[g * 4 + 3];
    const double md = metdet[g], rmd = rmetdet[g];
    for (int t = w; t < ntiles; t += nw) {  // wave-uniform trip count
      const int lev = t * 4 + sub;
      const bool live = lev < nlevels;
      const size_t o = ((size_t)e * nlevels + lev) * 16 + pt;
      // every lane runs the DPP code (rows of a dead level just compute on zeros)
      if (which == 0) {
        const double s = live ? __builtin_nontemporal_load(in + o) : 0.0;
        double g0, g1;
        gradient_sphere(c, Di, rrearth, s, g0, g1);
        dbl2 r;
        r.x = g0;
        r.y = g1;
        if (live) __builtin_nontemporal_store(r, reinterpret_cast<dbl2*>(out) + o);
      } else {
        dbl2 uv;
        uv.x = uv.y = 0.0;
        if (live) uv = __builtin_nontemporal_load(reinterpret_cast<const dbl2*>(in) + o);
        const double r = which == 1 ? divergence_sphere(c, Di, md, rmd, rrearth, uv.x, uv.y)
                                    : vorticity_sphere(c, Dm, rmd, rrearth, uv.x, uv.y);
        if (live) __builtin_nontemporal_store(r, out + o);
      }
    }
  }
}

__global__ __launch_bounds__(256) void sphere_operator_np8(int which, const double* __restrict__ in, double* __restrict__ out,
                                    const double* __restrict__ D, const double* __restrict__ Dinv,
                                    const double* __restrict__ metdet, const double* __restrict__ rmetdet,
                                    const double* __restrict__ dvv, int ie0, int ne, int nlevels, double rrearth) {
  constexpr int NP = np8::NP;
  __shared__ __attribute__((aligned(16))) double s_tile[4 * 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
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
  for (int e = blockIdx.x; e < ne; e += gridDim.x) {
    const size_t ie = (size_t)ie0 + e;
    const size_t g = ie * 64 + lane;
    const np8::M22 Di = np8::load_m22(Dinv + ie * 256, lane), Dm = np8::load_m22(D + ie * 256, lane);
    const double md = metdet[g], rmd = rmetdet[g];
    for (int lev = w; lev < nlevels; lev += nw) {  // wave-uniform
      const size_t o = ((size_t)e * nlevels + lev) * 64 + lane;
      np8::wave_lds_fence();  // the previous field's reads of the wave's tile are done
      if (which == 0) {
        double g0, g1;
        np8::gradient_sphere(c, lane, Di, rrearth, __builtin_nontemporal_load(in + o), g0, g1);
        dbl2 r;
        r.x = g0;
        r.y = g1;
        __builtin_nontemporal_store(r, reinterpret_cast<dbl2*>(out) + o);
      } else {
        const dbl2 uv = __builtin_nontemporal_load(reinterpret_cast<const dbl2*>(in) + o);
        const double r = which == 1 ? np8::divergence_sphere(c, lane, Di, md, rmd, rrearth, uv.x, uv.y)
                                    : np8::vorticity_sphere(c, lane, Dm, rmd, rrearth, uv.x, uv.y);
        __builtin_nontemporal_store(r, out + o);
      }
    }
  }
}

// The two vertical integrals as functions of their own (reference: compute_and_apply_rhs.hpp:11-17 declares them
// next to compute_and_apply_rhs; P:280-312 preq_hydrostatic, P:314-352 preq_omega_ps).  One thread per column, the
// reference's own operand order, FP contraction off and IEEE division, so the result is bit-identical to the
// reference's (the fused kernels use blocked scans and a Newton reciprocal instead: DESIGN.md "Numerics").
__global__ void preq_hydrostatic_kernel(int pp, int nlev, size_t ncols, const double* __restrict__ phis,
                                        const double* __restrict__ Tv, const double* __restrict__ p,
                                        const double* __restrict__ dp, double Rgas, double* __restrict__ phi) {
#pragma clang fp contract(off)
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += (size_t)gridDim.x * blockDim.x) {
    const size_t e = c / pp, q = c % pp, base = e * (size_t)nlev * pp + q;
    const double ph = phis[c];
    double phii = 0.0;  // P:298,305: phii[ilev+1]
    for (int k = nlev - 1; k >= 0; --k) {
      const size_t o = base + (size_t)k * pp;
      const double hkk = 0.5 * dp[o] / p[o];  // P:291,300,308
      const double hkl = 2.0 * hkk;           // P:292,301
      const double rt = Rgas * Tv[o];
      // P:294 (bottom level: no phii term), P:303, P:309
      phi[o] = k == nlev - 1 ? ph + rt * hkk : ph + phii + rt * hkk;
      phii = k == nlev - 1 ? rt * hkl : phii + rt * hkl;  // P:293,302
    }
  }
}
__global__ void preq_omega_ps_kernel(int pp, int nlev, size_t ncols, const double* __restrict__ p,
                                     const double* __restrict__ vgrad_p, const double* __restrict__ divdp,
                                     double* __restrict__ omega_p) {
#pragma clang fp contract(off)
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < ncols; c += (size_t)gridDim.x * blockDim.x) {
    const size_t e = c / pp, q = c % pp, base = e * (size_t)nlev * pp + q;
    double suml = 0.0;
    for (int k = 0; k < nlev; ++k) {
      const size_t o = base + (size_t)k * pp;
      const double ckk = 0.5 / p[o];  // P:323,333,344
      const double ckl = 2.0 * ckk;   // P:334,345
      const double term = divdp[o];
      // P:325 (top level: no suml term), P:336-337, P:347-348
      omega_p[o] = k == 0 ? vgrad_p[o] / p[o] - ckk * term : vgrad_p[o] / p[o] - ckl * suml - ckk * term;
      suml = k == 0 ? term : suml + term;  // P:326,339
    }
  }
}
hipError_t launch_preq_hydrostatic(int np, int nlev, int nelem, const double* phis, const double* Tv, const double* p,
                                   const double* dp, double Rgas, double* phi, hipStream_t s) {
  const size_t ncols = (size_t)nelem * np * np;
  if (ncols == 0) return hipSuccess;
  const size_t want = (ncols + 63) / 64;
  hipLaunchKernelGGL(preq_hydrostatic_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(64), 0, s, np * np, nlev,
                     ncols, phis, Tv, p, dp, Rgas, phi);
  return hipGetLastError();
}
hipError_t launch_preq_omega_ps(int np, int nlev, int nelem, const double* p, const double* vgrad_p, const double* divdp,
                                double* omega_p, hipStream_t s) {
  const size_t ncols = (size_t)nelem * np * np;
  if (ncols == 0) return hipSuccess;
  const size_t want = (ncols + 63) / 64;
  hipLaunchKernelGGL(preq_omega_ps_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(64), 0, s, np * np, nlev,
                     ncols, p, vgrad_p, divdp, omega_p);
  return hipGetLastError();
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
                                  const double* dvv, int ie, int ne, int nlevels, double rrearth, hipStream_t s) {
  if (nlevels <= 0 || ne <= 0) return hipSuccess;
  const unsigned grid = ne < 65536 ? ne : 65536;  // elements beyond that: grid-stride
  if (np == 4) {
    hipLaunchKernelGGL(sphere_operator_np4, dim3(grid), dim3(256), 0, s, which, in, out, D, Dinv, metdet, rmetdet,
                       dvv, ie, ne, nlevels, rrearth);
  } else if (np == 8) {
    hipLaunchKernelGGL(sphere_operator_np8, dim3(grid), dim3(256), 0, s, which, in, out, D, Dinv, metdet,
                       rmetdet, dvv, ie, ne, nlevels, rrearth);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace caar
