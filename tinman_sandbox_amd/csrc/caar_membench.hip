// caar_membench.hip — measurement utilities for the roofline (not on the product path).
//
//  * stream_copy: plain device copy, 8 or 16 bytes per lane, grid-stride — the
//    measured HBM ceiling quoted next to the 8 TB/s spec peak in bench.py/DESIGN.md and
//    the calibration run for the FETCH_SIZE/WRITE_SIZE counters
//    (/opt/skills/guides/MI355X_MICROARCH.md, HBM section: calibrate on a known byte count
//    in your own access pattern).
//  * traffic_skeleton: touches exactly the bytes compute_and_apply_rhs touches for NP=4
//    (same arrays, same tile/lane addressing, same load/store widths) with no
//    arithmetic worth mentioning: the bandwidth this access pattern can reach when
//    nothing but memory limits it.
#include <hip/hip_runtime.h>

#include "caar_kernel_args.h"

namespace caar {

template <typename V>
__global__ void stream_copy_kernel(V* __restrict__ dst, const V* __restrict__ src, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i];
}

hipError_t launch_stream_copy(double* dst, const double* src, size_t n_doubles, int lane_bytes,
                              hipStream_t stream) {
  const int threads = 256, blocks = 256 * 8;
  if (lane_bytes == 16 && n_doubles % 2 == 0)
    hipLaunchKernelGGL(stream_copy_kernel<double2>, dim3(blocks), dim3(threads), 0, stream,
                       reinterpret_cast<double2*>(dst), reinterpret_cast<const double2*>(src), n_doubles / 2);
  else
    hipLaunchKernelGGL(stream_copy_kernel<double>, dim3(blocks), dim3(threads), 0, stream, dst, src, n_doubles);
  return hipGetLastError();
}

template <int NLEV, int TPW>
__global__ __launch_bounds__(NLEV / 4 / TPW * 64) void traffic_skeleton_np4(const KernelArgs k) {
  constexpr int PP = 16, NT = NLEV / 4, BLK = NLEV * PP;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, pt = lane & 15;
  const size_t ie = (size_t)k.nets + blockIdx.x, tl = (size_t)k.timelevels;
  const double* dp_n0 = k.dp3d + (ie * tl + k.n0) * BLK;
  const double2* v_n0 = reinterpret_cast<const double2*>(k.v + (ie * tl + k.n0) * BLK * 2);
  const double* T_n0 = k.T + (ie * tl + k.n0) * BLK;
  const double* Qdp = k.Qdp + ((ie * k.qsize_d + 0) * 2 + (k.qn0 >= 0 ? k.qn0 : 0)) * BLK;
  const double2* v_nm1 = reinterpret_cast<const double2*>(k.v + (ie * tl + k.nm1) * BLK * 2);
  const double* T_nm1 = k.T + (ie * tl + k.nm1) * BLK;
  const double* dp_nm1 = k.dp3d + (ie * tl + k.nm1) * BLK;
  double2* v_np1 = reinterpret_cast<double2*>(k.v + (ie * tl + k.np1) * BLK * 2);
  double* T_np1 = k.T + (ie * tl + k.np1) * BLK;
  double* dp_np1 = k.dp3d + (ie * tl + k.np1) * BLK;
  double2* vn0 = reinterpret_cast<double2*>(k.vn0 + ie * BLK * 2);
  double* omega_p = k.omega_p + ie * BLK;
  double* phi = k.phi + ie * BLK;
  const double* pecnd = k.pecnd + ie * BLK;
  double* eta = k.eta_dot_dpdn + ie * (BLK + PP);
  // 13 metric values per point, read by every lane like the real kernel's LDS staging source
  double g = k.fcor[ie * PP + pt] + k.spheremp[ie * PP + pt] + k.metdet[ie * PP + pt] +
             k.rmetdet[ie * PP + pt] + k.phis[ie * PP + pt];
#pragma unroll
  for (int j = 0; j < 4; ++j) g += k.D[(ie * PP + pt) * 4 + j] + k.Dinv[(ie * PP + pt) * 4 + j];
  g *= 0.0;
#pragma unroll
  for (int r = 0; r < TPW; ++r) {
    const int off = (w * TPW + r) * 64 + lane;
    const double a = dp_n0[off] + T_n0[off] + Qdp[off] + pecnd[off] + g;
    const double2 uv = v_n0[off], um = v_nm1[off], un = vn0[off];
    double2 o;
    o.x = uv.x + um.x;
    o.y = uv.y + um.y;
    v_np1[off] = o;
    T_np1[off] = T_nm1[off] + a;
    dp_np1[off] = dp_nm1[off] + a;
    phi[off] = a;
    omega_p[off] = omega_p[off] + a * 0.0;
    o.x = un.x + a * 0.0;
    o.y = un.y;
    vn0[off] = o;
    eta[off] = eta[off] + 0.0;
  }
  if (tid < PP) eta[BLK + tid] = eta[BLK + tid] + 0.0;
  (void)NT;
}

hipError_t launch_traffic_skeleton(const KernelArgs& k, int nlev, int num_elems, hipStream_t stream) {
  if (nlev == 72)
    hipLaunchKernelGGL((traffic_skeleton_np4<72, 3>), dim3(num_elems), dim3(384), 0, stream, k);
  else if (nlev == 128)
    hipLaunchKernelGGL((traffic_skeleton_np4<128, 4>), dim3(num_elems), dim3(512), 0, stream, k);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace caar
