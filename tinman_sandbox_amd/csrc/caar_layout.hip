// caar_layout.hip — Fortran-layout ingest / egress for the element arrays.
//
// A HOMME/E3SM host keeps the fields as Fortran arrays, first index fastest
// (compute_and_apply_rhs_test/fortran/element_state_mod.F90:17-23, element_mod.F90:69-121):
//     v(np,np,2,nlev,timelevels)   T, dp3d(np,np,nlev,timelevels)   Qdp(np,np,nlev,qsize_d,2)
//     phi, omega_p, pecnd(np,np,nlev)   vn0(np,np,2,nlev)   eta_dot_dpdn(np,np,nlev+1)
//     D, Dinv(np,np,2,2)   fcor, spheremp, metdet, rmetdet, phis(np,np)
// each with a trailing element index once copied out of elem(ie) into flat arrays (what
// the reference's level_vectorized_ppscan/Elements.cpp:154-435 pull/push functions take).
// The kernels here convert between those flat Fortran-ordered arrays and the C++
// element-major layout of include/caar.h.  Logical indices coincide — C++ [a][b][c] is
// Fortran (a+1,b+1,c+1) — so for every array the outer order
// (element, time level, level) is the same on both sides and the conversion is a
// permutation INSIDE each np x np x ncomp block (128 B ... 2 KiB); only Qdp with
// qsize_d > 1 also permutes outer indices.  One thread per double, destination index
// linear (coalesced stores), source inside the same block (same cache lines): pure
// HBM-bandwidth kernels.
#include <hip/hip_runtime.h>

namespace caar {

// NP: points per edge; NC: components per point (1, 2, or 4 = the 2x2 metric tensors).
// Block-local offset of logical (a, b, comp) in each layout:
//   C++     ((a*NP + b)*NC + cc),  cc = c           (NC=2)   or r*2 + c   (NC=4, [r][c])
//   Fortran  a + NP*(b + NP*cf),   cf = c           (NC=2)   or r + 2*c   (NC=4, (r,c))
template <int NP, int NC, bool TO_CAAR>
__global__ void layout_kernel(double* __restrict__ dst, const double* __restrict__ src, size_t n,
                              int nlev, int qd, int qdp_outer) {
  constexpr int INNER = NP * NP * NC;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
       idx += (size_t)gridDim.x * blockDim.x) {
    const size_t outer = idx / INNER;
    const int inner = (int)(idx % INNER);
    int a, b, cc_c, cf;
    if (TO_CAAR) {  // destination block is in C++ order
      cc_c = inner % NC;
      b = (inner / NC) % NP;
      a = inner / (NC * NP);
      cf = NC == 4 ? (cc_c >> 1) + 2 * (cc_c & 1) : cc_c;
    } else {  // destination block is in Fortran order
      a = inner % NP;
      b = (inner / NP) % NP;
      cf = inner / (NP * NP);
      cc_c = NC == 4 ? (cf & 1) * 2 + (cf >> 1) : cf;
    }
    const int in_c = (a * NP + b) * NC + cc_c;
    const int in_f = a + NP * (b + NP * cf);
    size_t src_outer = outer;
    if (qdp_outer && qd > 1) {
      // C++ [ie][q][t][k]  <->  Fortran (k, q, t, ie): k fastest on both sides
      const int k = (int)(outer % nlev);
      size_t rest = outer / nlev;
      if (TO_CAAR) {  // dst outer = ((ie*qd + q)*2 + t)*nlev + k
        const int t = (int)(rest % 2);
        rest /= 2;
        const int q = (int)(rest % qd);
        const size_t ie = rest / qd;
        src_outer = ((ie * 2 + t) * qd + q) * nlev + k;
      } else {  // dst outer = ((ie*2 + t)*qd + q)*nlev + k
        const int q = (int)(rest % qd);
        rest /= qd;
        const int t = (int)(rest % 2);
        const size_t ie = rest / 2;
        src_outer = ((ie * qd + q) * 2 + t) * nlev + k;
      }
    }
    dst[idx] = src[src_outer * INNER + (TO_CAAR ? in_f : in_c)];
  }
}

// All arrays of one conversion in ONE launch (caar_layout_from_f90 / caar_layout_to_f90): sixteen back-to-back launches of
// 0.1-300 MB each pay a launch drain + fill per array (~10 % of a 0.7 ms conversion at 10 000 elements).  A workgroup
// owns one 4 KiB chunk (512 doubles) of one array's destination — the smallest unit of work is what the memory system
// likes best (the two-stream copy peaks with one 4 KiB chunk per workgroup) —, both of a lane's
// loads are requested before its first store, every access non-temporal (each byte is touched once).
struct LayoutJob {
  double* dst;
  const double* src;
  size_t n;              // doubles
  int nc, qdp_outer;
  unsigned first_block;  // this job's first workgroup in the grid
};
struct LayoutJobs {
  LayoutJob j[16];
  int count, nlev, qd;
};
constexpr int kLayoutChunk = 512;  // doubles per workgroup

template <int NP, int NC, bool TO_CAAR>
__device__ __forceinline__ size_t layout_source_index(size_t idx, int nlev, int qd, int qdp_outer) {
  constexpr int INNER = NP * NP * NC;
  const size_t outer = idx / INNER;
  const int inner = (int)(idx % INNER);
  int a, b, cc_c, cf;
  if (TO_CAAR) {
    cc_c = inner % NC;
    b = (inner / NC) % NP;
    a = inner / (NC * NP);
    cf = NC == 4 ? (cc_c >> 1) + 2 * (cc_c & 1) : cc_c;
  } else {
    a = inner % NP;
    b = (inner / NP) % NP;
    cf = inner / (NP * NP);
    cc_c = NC == 4 ? (cf & 1) * 2 + (cf >> 1) : cf;
  }
  const int in_c = (a * NP + b) * NC + cc_c;
  const int in_f = a + NP * (b + NP * cf);
  size_t src_outer = outer;
  if (qdp_outer && qd > 1) {  // as in layout_kernel
    const int k = (int)(outer % nlev);
    size_t rest = outer / nlev;
    if (TO_CAAR) {
      const int t = (int)(rest % 2);
      rest /= 2;
      const int q = (int)(rest % qd);
      const size_t ie = rest / qd;
      src_outer = ((ie * 2 + t) * qd + q) * nlev + k;
    } else {
      const int q = (int)(rest % qd);
      rest /= qd;
      const int t = (int)(rest % 2);
      const size_t ie = rest / 2;
      src_outer = ((ie * qd + q) * 2 + t) * nlev + k;
    }
  }
  return src_outer * INNER + (TO_CAAR ? in_f : in_c);
}

template <int NP, int NC, bool TO_CAAR>
__device__ __forceinline__ void layout_chunk(const LayoutJob& job, size_t base, int nlev, int qd) {
  const size_t i0 = base + threadIdx.x, i1 = i0 + 256;
  double x0 = 0, x1 = 0;
  if (i0 < job.n) x0 = __builtin_nontemporal_load(job.src + layout_source_index<NP, NC, TO_CAAR>(i0, nlev, qd, job.qdp_outer));
  if (i1 < job.n) x1 = __builtin_nontemporal_load(job.src + layout_source_index<NP, NC, TO_CAAR>(i1, nlev, qd, job.qdp_outer));
  if (i0 < job.n) __builtin_nontemporal_store(x0, job.dst + i0);
  if (i1 < job.n) __builtin_nontemporal_store(x1, job.dst + i1);
}

template <int NP, bool TO_CAAR>
__global__ __launch_bounds__(256) void layout_all_kernel(const LayoutJobs J) {
  int ji = 0;
  while (ji + 1 < J.count && blockIdx.x >= J.j[ji + 1].first_block) ++ji;  // uniform for the workgroup
  const LayoutJob& job = J.j[ji];
  const size_t base = (size_t)(blockIdx.x - job.first_block) * kLayoutChunk;
  if (job.nc == 1) layout_chunk<NP, 1, TO_CAAR>(job, base, J.nlev, J.qd);
  else if (job.nc == 2) layout_chunk<NP, 2, TO_CAAR>(job, base, J.nlev, J.qd);
  else layout_chunk<NP, 4, TO_CAAR>(job, base, J.nlev, J.qd);
}

// `count` arrays (dst/src already offset to the first element of the range) in one launch.
hipError_t launch_layout_all(int np, int count, double* const* dst, const double* const* src, const size_t* n, const int* nc,
                             const int* qdp_outer, int nlev, int qd, bool to_caar, hipStream_t s) {
  if (count < 0 || count > 16 || (np != 4 && np != 8)) return hipErrorInvalidValue;
  LayoutJobs J;
  J.count = 0;
  J.nlev = nlev;
  J.qd = qd;
  size_t blocks = 0;
  for (int i = 0; i < count; ++i) {
    if (n[i] == 0) continue;
    LayoutJob& j = J.j[J.count++];
    j.dst = dst[i];
    j.src = src[i];
    j.n = n[i];
    j.nc = nc[i];
    j.qdp_outer = qdp_outer[i];
    j.first_block = (unsigned)blocks;
    blocks += (n[i] + kLayoutChunk - 1) / kLayoutChunk;
    if (blocks > 0x7fffffffu) return hipErrorInvalidValue;
  }
  if (J.count == 0) return hipSuccess;
  const dim3 grid((unsigned)blocks), block(256);
  if (np == 4) {
    if (to_caar) hipLaunchKernelGGL((layout_all_kernel<4, true>), grid, block, 0, s, J);
    else hipLaunchKernelGGL((layout_all_kernel<4, false>), grid, block, 0, s, J);
  } else {
    if (to_caar) hipLaunchKernelGGL((layout_all_kernel<8, true>), grid, block, 0, s, J);
    else hipLaunchKernelGGL((layout_all_kernel<8, false>), grid, block, 0, s, J);
  }
  return hipGetLastError();
}

template <int NP, bool TO_CAAR>
static hipError_t launch_np(double* dst, const double* src, size_t n, int nc, int nlev, int qd, int qdp_outer,
                            hipStream_t s) {
  if (n == 0) return hipSuccess;
  const int threads = 256;
  size_t want = (n + threads - 1) / threads;
  const int blocks = (int)(want < 256 * 16 ? want : 256 * 16);
  if (nc == 1) hipLaunchKernelGGL((layout_kernel<NP, 1, TO_CAAR>), dim3(blocks), dim3(threads), 0, s, dst, src, n, nlev, qd, qdp_outer);
  else if (nc == 2) hipLaunchKernelGGL((layout_kernel<NP, 2, TO_CAAR>), dim3(blocks), dim3(threads), 0, s, dst, src, n, nlev, qd, qdp_outer);
  else hipLaunchKernelGGL((layout_kernel<NP, 4, TO_CAAR>), dim3(blocks), dim3(threads), 0, s, dst, src, n, nlev, qd, qdp_outer);
  return hipGetLastError();
}

// One array: n doubles starting at dst/src (already offset to the first element of the range).
hipError_t launch_layout(double* dst, const double* src, size_t n, int np, int nc, int nlev, int qd,
                         int qdp_outer, bool to_caar, hipStream_t s) {
  if (np == 4) return to_caar ? launch_np<4, true>(dst, src, n, nc, nlev, qd, qdp_outer, s)
                              : launch_np<4, false>(dst, src, n, nc, nlev, qd, qdp_outer, s);
  if (np == 8) return to_caar ? launch_np<8, true>(dst, src, n, nc, nlev, qd, qdp_outer, s)
                              : launch_np<8, false>(dst, src, n, nc, nlev, qd, qdp_outer, s);
  return hipErrorInvalidValue;
}

}  // namespace caar
