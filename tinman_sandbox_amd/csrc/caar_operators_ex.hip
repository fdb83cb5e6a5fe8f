// caar_operators_ex.hip — the sphere operators NEXT TO the CAAR path as device entry points.
//
// Reference: cxx/level_vectorized_ppscan/SphereOperators.hpp (K:) — gradient_sphere_update K:271-312,
// divergence_sphere_update K:363-403, vorticity_sphere_vector K:452-490, divergence_sphere_wk K:494-534,
// laplace_simple K:538-550, laplace_tensor K:556-596, laplace_tensor_replace K:600-637, curl_sphere_wk_testcov K:640-690,
// grad_sphere_wk_testcov K:694-770, vlaplace_sphere_wk_cartesian(_reduced) K:777-915,
// vlaplace_sphere_wk_contra K:938-993.  Parity unpinned (the reference never builds or calls them): see
// oracle/sphere_ops_oracle.c and tests/test_sphere_ops.py for what pins the oracle they are tested against.
//
// Every one of them is pointwise algebra around the same four NP x NP contractions of one level,
//     da(f)[a][b] = sum_k Dvv[k][a] f[k][b]      db(f)[a][b] = sum_k Dvv[k][b] f[a][k]     (strong forms)
//     wa(f)[a][b] = sum_k Dvv[a][k] f[k][b]      wb(f)[a][b] = sum_k Dvv[b][k] f[a][k]     (weak forms: Dvv^T)
// in this repository's index convention (field[a][b] == Fortran (a+1, b+1); K: uses the transposed one, the
// translation is spelled out in oracle/sphere_ops_oracle.c).  They run on the building blocks of the fused
// kernels: NP=4 one level = one 16-lane DPP row (caar_np4_ops.h: row_ror / quad_perm moves, no LDS), NP=8 one
// level = one wavefront on v_mfma_f64_4x4x4 (caar_np8_ops.h "MFMA form"); the weak forms are the same code with
// the coefficient slices taken from Dvv^T.  Bandwidth-bound: every input byte read once, every output written once.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "../../include/caar.h"
#include "caar_kernel_args.h"
#include "caar_np4_ops.h"
#include "caar_np8_ops.h"

namespace caar {

// contraction providers ---------------------------------------------------------------------------
struct X4 {  // NP=4: lane = sub*16 + a*4 + b
  RowCoef c, ct;
  __device__ __forceinline__ double da(double f) const { return d_da(c, f); }
  __device__ __forceinline__ double db(double f) const { return d_db(c, f); }
  __device__ __forceinline__ double wa(double f) const { return d_da(ct, f); }
  __device__ __forceinline__ double wb(double f) const { return d_db(ct, f); }
};
struct X4M {  // NP=4 on the matrix cores: lane = 16a + 4lev + b (caar_np4_ops.h "MFMA form")
  Mfma4Ctx c;
  __device__ __forceinline__ double da(double f) const { return mfma4_d_da(c, f); }
  __device__ __forceinline__ double db(double f) const { return mfma4_d_db(c, f); }
  __device__ __forceinline__ double wa(double f) const { return mfma4_w_a(c, f); }
  __device__ __forceinline__ double wb(double f) const { return mfma4_w_b(c, f); }
};
struct X8 {  // NP=8: lane = MFMA result layout (np8::mfma_point)
  np8::MfmaCtx c, ct;
  __device__ __forceinline__ double da(double f) const { return np8::mfma_d_da(c, f); }
  __device__ __forceinline__ double db(double f) const { return np8::mfma_d_db(c, f); }
  __device__ __forceinline__ double wa(double f) const { return np8::mfma_d_da(ct, f); }
  __device__ __forceinline__ double wb(double f) const { return np8::mfma_d_db(ct, f); }
};

struct T22 {  // [r][c] of this lane's point
  double m00, m01, m10, m11;
};
__device__ __forceinline__ T22 load_t22(const double* base, size_t point) {
  const dbl2* p = reinterpret_cast<const dbl2*>(base + point * 4);
  const dbl2 r0 = p[0], r1 = p[1];
  T22 t;
  t.m00 = r0.x; t.m01 = r0.y; t.m10 = r1.x; t.m11 = r1.y;
  return t;
}
struct V2 {
  double x, y;
};

// the operators (one level, this lane's point) ----------------------------------------------------
// S:9-48 == K:229-269
template <class X>
__device__ __forceinline__ V2 op_gradient(const X& x, const T22& Dinv, double rr, double s) {
  const double v1 = x.da(s) * rr, v2 = x.db(s) * rr;
  return {Dinv.m00 * v1 + Dinv.m10 * v2, Dinv.m01 * v1 + Dinv.m11 * v2};
}
// S:50-89 == K:315-358
template <class X>
__device__ __forceinline__ double op_divergence(const X& x, const T22& Dinv, double metdet, double rmetdet, double rr, V2 v) {
  const double gv0 = metdet * (Dinv.m00 * v.x + Dinv.m01 * v.y);
  const double gv1 = metdet * (Dinv.m10 * v.x + Dinv.m11 * v.y);
  return (x.da(gv0) + x.db(gv1)) * rmetdet * rr;
}
// S:91-129 == K:452-490
template <class X>
__device__ __forceinline__ double op_vorticity(const X& x, const T22& D, double rmetdet, double rr, V2 v) {
  const double vc0 = D.m00 * v.x + D.m10 * v.y;
  const double vc1 = D.m01 * v.x + D.m11 * v.y;
  return (x.da(vc1) - x.db(vc0)) * rmetdet * rr;
}
// K:494-534: div(m,n) = -sum_j (spheremp(j,n) gv1(j,n) Dvv(m,j) + spheremp(m,j) gv2(m,j) Dvv(n,j)) rrearth
template <class X>
__device__ __forceinline__ double op_divergence_wk(const X& x, const T22& Dinv, double spheremp, double rr, V2 v) {
  const double gv0 = Dinv.m00 * v.x + Dinv.m01 * v.y;
  const double gv1 = Dinv.m10 * v.x + Dinv.m11 * v.y;
  return -(x.wa(spheremp * gv0) + x.wb(spheremp * gv1)) * rr;
}
// K:556-596 (TENSOR false: K:538-550 laplace_simple)
template <bool TENSOR, class X>
__device__ __forceinline__ double op_laplace(const X& x, const T22& Dinv, double spheremp, const T22& tv, double rr, double s) {
  V2 g = op_gradient(x, Dinv, rr, s);
  if (TENSOR) g = {tv.m00 * g.x + tv.m01 * g.y, tv.m10 * g.x + tv.m11 * g.y};
  return op_divergence_wk(x, Dinv, spheremp, rr, g);
}
// K:640-690
template <class X>
__device__ __forceinline__ V2 op_curl_wk_testcov(const X& x, const T22& D, double mp, double rr, double s) {
  const double ms = mp * s;
  const double c0 = -x.wb(ms), c1 = x.wa(ms);
  return {(D.m00 * c0 + D.m01 * c1) * rr, (D.m10 * c0 + D.m11 * c1) * rr};
}
// K:694-770
template <class X>
__device__ __forceinline__ V2 op_grad_wk_testcov(const X& x, const T22& D, double mp, const T22& metinv, double metdet,
                                                 double rr, double s) {
  const double ms = mp * s;
  const double A = x.wa(ms), B = x.wb(ms);
  const double c0 = -(metinv.m00 * metdet * A + metinv.m10 * metdet * B);
  const double c1 = -(metinv.m01 * metdet * A + metinv.m11 * metdet * B);
  return {(D.m00 * c0 + D.m01 * c1) * rr, (D.m10 * c0 + D.m11 * c1) * rr};
}

// which-codes of include/caar.h
enum {
  OP_GRAD = CAAR_OP_GRADIENT_SPHERE, OP_DIV = CAAR_OP_DIVERGENCE_SPHERE, OP_VORT = CAAR_OP_VORTICITY_SPHERE,
  OP_DIV_WK = CAAR_OP_DIVERGENCE_SPHERE_WK, OP_LAP = CAAR_OP_LAPLACE_SIMPLE, OP_LAP_T = CAAR_OP_LAPLACE_TENSOR,
  OP_CURL_WK = CAAR_OP_CURL_SPHERE_WK_TESTCOV, OP_GRAD_WK = CAAR_OP_GRAD_SPHERE_WK_TESTCOV,
  OP_VLAP_CONTRA = CAAR_OP_VLAPLACE_SPHERE_WK_CONTRA, OP_VLAP_CART = CAAR_OP_VLAPLACE_SPHERE_WK_CARTESIAN,
  OP_GRAD_UPD = CAAR_OP_GRADIENT_SPHERE_UPDATE, OP_DIV_UPD = CAAR_OP_DIVERGENCE_SPHERE_UPDATE,
  OP_VLAP_CART_DAMPED = CAAR_OP_VLAPLACE_SPHERE_WK_CARTESIAN_DAMPED,
  // K:600-637 "a version of laplace_tensor where input is replaced by output": every lane reads its own point of a.out,
  // the NP x NP exchange happens in registers, and the same lane stores the result to the same place
  OP_LAP_T_REPL = CAAR_OP_LAPLACE_TENSOR_REPLACE
};
__host__ __device__ constexpr bool op_vector_in(int w) {
  return w == OP_DIV || w == OP_VORT || w == OP_DIV_WK || w == OP_VLAP_CONTRA || w == OP_VLAP_CART ||
         w == OP_DIV_UPD || w == OP_VLAP_CART_DAMPED;
}
__host__ __device__ constexpr bool op_vector_out(int w) {
  return w == OP_GRAD || w == OP_CURL_WK || w == OP_GRAD_WK || w == OP_VLAP_CONTRA || w == OP_VLAP_CART ||
         w == OP_GRAD_UPD || w == OP_VLAP_CART_DAMPED;
}

// One workgroup per element (grid-stride), 4 waves; NP=4: a wave walks tiles of 4 levels, NP=8: levels.
template <int NP, int WHICH>
__global__ __launch_bounds__(256) void sphere_operator_ex_kernel(const OpArgs a) {
  constexpr int PP = NP * NP;
  __shared__ double s_dvv[PP], s_dvvT[PP];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
  for (int i = tid; i < PP; i += blockDim.x) {
    const double d = a.dvv[i];
    s_dvv[i] = d;
    s_dvvT[(i % NP) * NP + i / NP] = d;
  }
  __syncthreads();
  // The composites (laplace_*: 2 contraction pairs per point, vlaplace_*: 4-6, for 1-2 KiB of traffic per tile) are
  // bound by VALU issue in the DPP form (14 cross-lane moves + 8 FMAs per pair: 4.1-4.9 TB/s at NP=4,
  // profiles/r03/operator_bench_large.log); they run their contractions on the matrix cores instead.
  constexpr bool COMPOSITE = WHICH == OP_LAP || WHICH == OP_LAP_T || WHICH == OP_LAP_T_REPL || WHICH == OP_VLAP_CONTRA ||
                             WHICH == OP_VLAP_CART || WHICH == OP_VLAP_CART_DAMPED;
  constexpr bool MF4 = NP == 4 && COMPOSITE;
  typename std::conditional<NP == 4, typename std::conditional<MF4, X4M, X4>::type, X8>::type x;
  int pt, sub;
  if constexpr (MF4) {
    x.c = make_mfma4_ctx(s_dvv, lane);
    pt = mfma4_point(lane);
    sub = mfma4_level(lane);
  } else if constexpr (NP == 4) {
    x.c = make_row_coef(s_dvv, lane);
    x.ct = make_row_coef(s_dvvT, lane);
    pt = lane & 15;
    sub = lane >> 4;
  } else {
    x.c = np8::make_mfma_ctx(s_dvv, lane);
    x.ct = np8::make_mfma_ctx(s_dvvT, lane);
    pt = np8::mfma_point(lane);
    sub = 0;
  }
  constexpr int LPT = NP == 4 ? 4 : 1;  // levels per wave step
  const int nsteps = (a.nlevels + LPT - 1) / LPT;
  const double rr = a.rrearth;

  constexpr bool NEED_D = WHICH == OP_VORT || WHICH == OP_CURL_WK || WHICH == OP_GRAD_WK || WHICH == OP_VLAP_CONTRA;
  constexpr bool IN_PLACE = WHICH == OP_LAP_T_REPL;
  constexpr bool NEED_DINV = WHICH != OP_VORT && WHICH != OP_CURL_WK && WHICH != OP_GRAD_WK;
  constexpr bool NEED_METDET = WHICH == OP_DIV || WHICH == OP_GRAD_WK || WHICH == OP_VLAP_CONTRA || WHICH == OP_DIV_UPD;
  constexpr bool NEED_RMETDET = WHICH == OP_DIV || WHICH == OP_VORT || WHICH == OP_VLAP_CONTRA || WHICH == OP_DIV_UPD;
  constexpr bool NEED_SPHEREMP = WHICH == OP_DIV_WK || WHICH == OP_LAP || WHICH == OP_LAP_T || WHICH == OP_LAP_T_REPL || WHICH == OP_VLAP_CONTRA ||
                                 WHICH == OP_VLAP_CART || WHICH == OP_VLAP_CART_DAMPED;
  constexpr bool NEED_MP = WHICH == OP_CURL_WK || WHICH == OP_GRAD_WK || WHICH == OP_VLAP_CONTRA;
  constexpr bool NEED_METINV = WHICH == OP_GRAD_WK || WHICH == OP_VLAP_CONTRA;
  constexpr bool NEED_TV = WHICH == OP_LAP_T || WHICH == OP_LAP_T_REPL || WHICH == OP_VLAP_CART || WHICH == OP_VLAP_CART_DAMPED;
  constexpr bool NEED_S2C = WHICH == OP_VLAP_CART || WHICH == OP_VLAP_CART_DAMPED;

  for (int e = blockIdx.x; e < a.ne; e += gridDim.x) {
    const size_t g = (size_t)(a.e0 + e) * PP + pt;
    T22 D = {}, Dinv = {}, metinv = {}, tv = {};
    double metdet = 0, rmetdet = 0, spheremp = 0, mp = 0, s2c[3][2] = {};
    if (NEED_D) D = load_t22(a.D, g);
    if (NEED_DINV) Dinv = load_t22(a.Dinv, g);
    if (NEED_METINV) metinv = load_t22(a.metinv, g);
    if (NEED_TV) tv = load_t22(a.tensorVisc, g);
    if (NEED_METDET) metdet = a.metdet[g];
    if (NEED_RMETDET) rmetdet = a.rmetdet[g];
    if (NEED_SPHEREMP) spheremp = a.spheremp[g];
    if (NEED_MP) mp = a.mp[g];
    if (NEED_S2C) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const dbl2 t = *reinterpret_cast<const dbl2*>(a.vec_sph2cart + (g * 3 + k) * 2);
        s2c[k][0] = t.x;
        s2c[k][1] = t.y;
      }
    }
    // PREFETCH: the input of the wave's NEXT tile is requested before the current one is computed (+1..4 % on the
    // memory-bound operators; profiles/r03/operator_bench_large.log: with / without).
    constexpr bool PREFETCH = true;
    auto load_in = [&](int st_, double& s_, V2& v_, V2& old_) {
      const int lev_ = st_ * LPT + sub;
      s_ = 0;
      v_ = {0, 0};
      old_ = {0, 0};
      if (st_ < nsteps && lev_ < a.nlevels) {
        const size_t o_ = ((size_t)e * a.nlevels + lev_) * PP + pt;
        if (WHICH == OP_GRAD_UPD) {  // the accumulated-into values of the update forms travel with the input
          const dbl2 t = *(reinterpret_cast<const dbl2*>(a.out) + o_);
          old_ = {t.x, t.y};
        } else if (WHICH == OP_DIV_UPD) {
          old_.x = a.out[o_];
        }
        if (op_vector_in(WHICH)) {
          const dbl2 t = __builtin_nontemporal_load(reinterpret_cast<const dbl2*>(a.in) + o_);
          v_ = {t.x, t.y};
        } else {
          s_ = __builtin_nontemporal_load((IN_PLACE ? a.out : a.in) + o_);
        }
      }
    };
    double s_next = 0;
    V2 v_next = {0, 0}, old_next = {0, 0};
    if (PREFETCH) load_in(w, s_next, v_next, old_next);
    for (int st = w; st < nsteps; st += nw) {  // wave-uniform trip count; every lane runs the cross-lane code
      const int lev = st * LPT + sub;
      const bool live = lev < a.nlevels;
      const size_t o = ((size_t)e * a.nlevels + lev) * PP + pt;
      if (!PREFETCH) load_in(st, s_next, v_next, old_next);
      const double s = s_next;
      const V2 v = v_next, old = old_next;
      if (PREFETCH) load_in(st + nw, s_next, v_next, old_next);
      double rs = 0;
      V2 rv = {0, 0};
      if constexpr (WHICH == OP_GRAD || WHICH == OP_GRAD_UPD) rv = op_gradient(x, Dinv, rr, s);
      else if constexpr (WHICH == OP_DIV || WHICH == OP_DIV_UPD) rs = op_divergence(x, Dinv, metdet, rmetdet, rr, v);
      else if constexpr (WHICH == OP_VORT) rs = op_vorticity(x, D, rmetdet, rr, v);
      else if constexpr (WHICH == OP_DIV_WK) rs = op_divergence_wk(x, Dinv, spheremp, rr, v);
      else if constexpr (WHICH == OP_LAP) rs = op_laplace<false>(x, Dinv, spheremp, tv, rr, s);
      else if constexpr (WHICH == OP_LAP_T || WHICH == OP_LAP_T_REPL) rs = op_laplace<true>(x, Dinv, spheremp, tv, rr, s);
      else if constexpr (WHICH == OP_CURL_WK) rv = op_curl_wk_testcov(x, D, mp, rr, s);
      else if constexpr (WHICH == OP_GRAD_WK) rv = op_grad_wk_testcov(x, D, mp, metinv, metdet, rr, s);
      else if constexpr (WHICH == OP_VLAP_CONTRA) {  // K:938-993
        const double div = op_divergence(x, Dinv, metdet, rmetdet, rr, v) * a.nu_ratio;
        const double vort = op_vorticity(x, D, rmetdet, rr, v);
        const V2 gc = op_grad_wk_testcov(x, D, mp, metinv, metdet, rr, div);
        const V2 cc = op_curl_wk_testcov(x, D, mp, rr, vort);
        rv = {2.0 * spheremp * v.x * rr * rr + (gc.x - cc.x), 2.0 * spheremp * v.y * rr * rr + (gc.y - cc.y)};
      } else if constexpr (WHICH == OP_VLAP_CART || WHICH == OP_VLAP_CART_DAMPED) {  // K:849-915 / K:777-844
        double l[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) l[k] = op_laplace<true>(x, Dinv, spheremp, tv, rr, s2c[k][0] * v.x + s2c[k][1] * v.y);
        rv = {s2c[0][0] * l[0] + s2c[1][0] * l[1] + s2c[2][0] * l[2], s2c[0][1] * l[0] + s2c[1][1] * l[1] + s2c[2][1] * l[2]};
        if (WHICH == OP_VLAP_CART) {  // the rigid-rotation term of UNDAMPRRCART (K:891-897)
          rv.x += 2.0 * spheremp * v.x * rr * rr;
          rv.y += 2.0 * spheremp * v.y * rr * rr;
        }
      }
      if (!live) continue;
      if (op_vector_out(WHICH)) {
        dbl2* po = reinterpret_cast<dbl2*>(a.out) + o;
        dbl2 t = {rv.x, rv.y};
        if (WHICH == OP_GRAD_UPD) t = {old.x + rv.x, old.y + rv.y};  // K:303-308: grad_s += ...
        __builtin_nontemporal_store(t, po);
      } else {
        double t = rs;
        if (WHICH == OP_DIV_UPD) {  // K:398-399: div_v *= beta; div_v += alpha * (...)
          t = old.x * a.beta;
          t += a.alpha * rs;
        }
        __builtin_nontemporal_store(t, a.out + o);
      }
    }
  }
}

template <int NP>
static hipError_t launch_ex_np(int which, const OpArgs& a, hipStream_t s) {
  const dim3 grid(a.ne < 65536 ? a.ne : 65536), block(256);
#define CAAR_OP_CASE(W) \
  case W: hipLaunchKernelGGL((sphere_operator_ex_kernel<NP, W>), grid, block, 0, s, a); break;
  switch (which) {
    CAAR_OP_CASE(OP_GRAD) CAAR_OP_CASE(OP_DIV) CAAR_OP_CASE(OP_VORT) CAAR_OP_CASE(OP_DIV_WK) CAAR_OP_CASE(OP_LAP)
    CAAR_OP_CASE(OP_LAP_T) CAAR_OP_CASE(OP_CURL_WK) CAAR_OP_CASE(OP_GRAD_WK) CAAR_OP_CASE(OP_VLAP_CONTRA)
    CAAR_OP_CASE(OP_VLAP_CART) CAAR_OP_CASE(OP_GRAD_UPD) CAAR_OP_CASE(OP_DIV_UPD) CAAR_OP_CASE(OP_VLAP_CART_DAMPED)
    CAAR_OP_CASE(OP_LAP_T_REPL)
    default: return hipErrorInvalidValue;
  }
#undef CAAR_OP_CASE
  return hipGetLastError();
}

hipError_t launch_sphere_operator_ex(int np, int which, const OpArgs& a, hipStream_t s) {
  if (a.ne <= 0 || a.nlevels <= 0) return hipSuccess;
  if (np == 4) return launch_ex_np<4>(which, a, s);
  if (np == 8) return launch_ex_np<8>(which, a, s);
  return hipErrorInvalidValue;
}

// ---- the tracer step sketched in EulerStepFunctor.hpp:32-68 --------------------------------------------------
// What the functor states, per tracer q and level: vstar_qdp = vstar * Qdp(qn0, q) (E:59-60); qtens = Qdp(qn0, q)
// (E:61); divergence_sphere_update(alpha = -dt, beta = 1.0, v = vstar_qdp, div_v = qtens) (E:65-66), i.e.
// qtens = Qdp - dt * div(vstar * Qdp).  The reference cannot compile it (the call passes 8 arguments to the
// 9-parameter K:363-370); this kernel is that statement, fused: vstar of a level is read ONCE for all tracers,
// Qdp once, qtens written once, no vstar_qdp buffer.  (2 + 2 qsize) field blocks of traffic per element.
template <int NP>
__global__ __launch_bounds__(256) void euler_step_kernel(const EulerArgs a) {
  constexpr int PP = NP * NP;
  __shared__ double s_dvv[PP];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nw = blockDim.x >> 6;
  for (int i = tid; i < PP; i += blockDim.x) s_dvv[i] = a.dvv[i];
  __syncthreads();
  typename std::conditional<NP == 4, X4, X8>::type x;
  int pt, sub;
  if constexpr (NP == 4) {
    x.c = x.ct = make_row_coef(s_dvv, lane);
    pt = lane & 15;
    sub = lane >> 4;
  } else {
    x.c = x.ct = np8::make_mfma_ctx(s_dvv, lane);
    pt = np8::mfma_point(lane);
    sub = 0;
  }
  constexpr int LPT = NP == 4 ? 4 : 1;
  const int nsteps = (a.nlev + LPT - 1) / LPT;
  const size_t block = (size_t)a.nlev * PP;  // one field block
  for (int e = blockIdx.x; e < a.ne; e += gridDim.x) {
    const size_t g = (size_t)(a.e0 + e) * PP + pt;
    const T22 Dinv = load_t22(a.Dinv, g);
    const double metdet = a.metdet[g], rmetdet = a.rmetdet[g];
    const double* qdp_e = a.qdp + ((size_t)(a.e0 + e) * a.qsize_d * 2 + a.qn0) * block;  // + q * 2 * block
    double* qtens_e = a.qtens + (size_t)e * a.qsize * block;                                  // + q * block
    for (int st = w; st < nsteps; st += nw) {
      const int lev = st * LPT + sub;
      const bool live = lev < a.nlev;
      const size_t o = (size_t)lev * PP + pt;
      dbl2 vs = {0, 0};
      if (live) vs = __builtin_nontemporal_load(reinterpret_cast<const dbl2*>(a.vstar) + (size_t)e * block + o);
      double q_next = live && a.qsize > 0 ? __builtin_nontemporal_load(qdp_e + o) : 0.0;
      for (int q = 0; q < a.qsize; ++q) {  // wave-uniform trip count
        const double qdp = q_next;
        if (q + 1 < a.qsize && live) q_next = __builtin_nontemporal_load(qdp_e + (size_t)(q + 1) * 2 * block + o);
        const double div = op_divergence(x, Dinv, metdet, rmetdet, a.rrearth, V2{vs.x * qdp, vs.y * qdp});
        double t = qdp * 1.0;  // K:398 with beta = 1.0
        t += -a.dt * div;      // K:399 with alpha = -dt
        if (live) __builtin_nontemporal_store(t, qtens_e + (size_t)q * block + o);
      }
    }
  }
}

hipError_t launch_euler_step(int np, const EulerArgs& a, hipStream_t s) {
  if (a.ne <= 0 || a.qsize <= 0 || a.nlev <= 0) return hipSuccess;
  const dim3 grid(a.ne < 65536 ? a.ne : 65536), block(256);
  if (np == 4) hipLaunchKernelGGL((euler_step_kernel<4>), grid, block, 0, s, a);
  else if (np == 8) hipLaunchKernelGGL((euler_step_kernel<8>), grid, block, 0, s, a);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// which geometry arrays operator `which` reads (bit i = member i of CaarOperatorGeometry), for validation
unsigned sphere_operator_ex_needs(int which) {
  enum { gD = 1, gDinv = 2, gMetdet = 4, gRmetdet = 8, gSpheremp = 16, gMp = 32, gMetinv = 64, gTv = 128, gS2c = 256 };
  switch (which) {
    case OP_GRAD: case OP_GRAD_UPD: return gDinv;
    case OP_DIV: case OP_DIV_UPD: return gDinv | gMetdet | gRmetdet;
    case OP_VORT: return gD | gRmetdet;
    case OP_DIV_WK: case OP_LAP: return gDinv | gSpheremp;
    case OP_LAP_T: case OP_LAP_T_REPL: return gDinv | gSpheremp | gTv;
    case OP_CURL_WK: return gD | gMp;
    case OP_GRAD_WK: return gD | gMp | gMetinv | gMetdet;
    case OP_VLAP_CONTRA: return gD | gDinv | gMp | gSpheremp | gMetinv | gMetdet | gRmetdet;
    case OP_VLAP_CART: case OP_VLAP_CART_DAMPED: return gDinv | gSpheremp | gTv | gS2c;
    default: return ~0u;
  }
}

}  // namespace caar
