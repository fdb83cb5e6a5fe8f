// caar_kernel_args.h — kernel argument block shared by the HIP kernels and the ABI layer.
#ifndef CAAR_KERNEL_ARGS_H
#define CAAR_KERNEL_ARGS_H

#include <hip/hip_runtime.h>

namespace caar {

// Device pointers to element 0 of each array (layouts: include/caar.h) plus the
// scalars of Homme::Control / Constants / HVCoord (data_structures.hpp:10-76).
struct KernelArgs {
  const double* D;
  const double* Dinv;
  const double* fcor;
  const double* spheremp;
  const double* metdet;
  const double* rmetdet;
  double* dp3d;
  double* v;
  double* T;
  const double* phis;
  const double* Qdp;
  double* eta_dot_dpdn;
  double* omega_p;
  double* phi;
  const double* pecnd;
  double* vn0;
  const double* Dvv;  // np*np, row-major Dvv[i][j]
  int nets;           // first element of this launch; blockIdx.x counts from it
  int n0, np1, nm1;
  int qn0;            // -1: dry
  int qsize_d, timelevels;
  double dt2;
  double rrearth;
  double eta_ave_w;
  double rv_over_rd_m1;  // Rwater_vapor/Rgas - 1.0 (P:151), formed on the host in fp64
  double Rgas;
  double kappa;
  double p_top;          // hyai[0]*ps0 (P:84)
};

// 1/x for a normal, non-zero fp64 x: v_rcp_f64 seed + two Newton steps (5 instructions,
// <= 1 ulp), instead of the IEEE division sequence (v_div_scale/fmas/fixup, ~12
// instructions and twice the live registers).  The reference divides by p and dp3d
// (P:150,219,291,323); both are positive pressures, so no scaling or fix-up is needed.
// Error budget: DESIGN.md "Numerics".
__device__ __forceinline__ double recip(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}

// Streaming accesses.  Every element array is read once and written once per launch
// and never re-used by another workgroup, so the loads and stores carry the
// non-temporal hint ("nt"): measured +5..8 % bandwidth on the traffic skeleton when
// BOTH loads and stores are non-temporal (profiles/r01/kbench_skeleton_nt.log).
typedef double dbl2 __attribute__((ext_vector_type(2)));
template <bool NT, typename T>
__device__ __forceinline__ T stream_load(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT, typename T>
__device__ __forceinline__ void stream_store(T* p, T v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// One compiled kernel configuration for a given (np, nlev).
struct KernelVariant {
  const char* kernel;  // demangled kernel name as rocprofv3 prints it
  const char* what;
  hipError_t (*launch)(const KernelArgs&, int num_elems, hipStream_t stream);
};

}  // namespace caar
#endif
