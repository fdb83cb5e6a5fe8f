// caar_operators.hip — the two vertical integrals of the path as stand-alone device entry points
// (caar_preq_hydrostatic / caar_preq_omega_ps) and the reciprocal numerics hook.  The sphere operators
// as stand-alone launches (caar_sphere_operator, _range, _ex) live in caar_operators_ex.hip.
#include <hip/hip_runtime.h>

#include "caar_kernel_args.h"
#include "caar_np4_ops.h"
#include "caar_np8_ops.h"

namespace caar {

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

}  // namespace caar
