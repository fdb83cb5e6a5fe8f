// caar_norms.hip — device side of print_results_2norm
// (compute_and_apply_rhs_test/cxx/pointers_only/compute_and_apply_rhs.cpp:353-399,
//  fortran/utils_mod.F90:9-30): per element, Kahan-compensated sum of squares of
// v, T and dp3d at one time level.  One thread per (element, field) walks its
// block in the reference's order so the result is the reference's, bit for bit;
// this is a diagnostic, not a hot path.
#include <hip/hip_runtime.h>

namespace caar {

#pragma clang fp contract(off)
__device__ double kahan_sum_sq(const double* __restrict__ f, int n) {
  double norm = 0.0, c = 0.0;
  for (int i = 0; i < n; ++i) {
    const double y = f[i] * f[i] - c;  // P:362
    const double t = norm + y;         // P:363
    c = (t - norm) - y;                // P:364
    norm = t;
  }
  return norm;
}

__global__ void state_norms_kernel(const double* __restrict__ v, const double* __restrict__ T,
                                   const double* __restrict__ dp, int blk, int timelevels, int tl,
                                   int e0, int n_elems, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * n_elems) return;
  const int e = i / 3, f = i % 3;
  const size_t ie = (size_t)e0 + e;
  const size_t slab = ie * timelevels + tl;
  double s;
  if (f == 0) s = kahan_sum_sq(v + slab * blk * 2, blk * 2);
  else if (f == 1) s = kahan_sum_sq(T + slab * blk, blk);
  else s = kahan_sum_sq(dp + slab * blk, blk);
  // P:388: pow(compute_norm(...), 2) — the reference takes the root and squares it again
  const double r = sqrt(s);
  out[i] = r * r;
}

hipError_t launch_state_norms(const double* v, const double* T, const double* dp, int np, int nlev,
                              int timelevels, int tl, int e0, int e1, double* out, hipStream_t stream) {
  const int n = e1 - e0;
  if (n <= 0) return hipSuccess;
  const int threads = 64, blocks = (3 * n + threads - 1) / threads;
  hipLaunchKernelGGL(state_norms_kernel, dim3(blocks), dim3(threads), 0, stream, v, T, dp,
                     np * np * nlev, timelevels, tl, e0, n, out);
  return hipGetLastError();
}

}  // namespace caar
