// homme_data.cpp — allocation and the closed-form synthetic initialiser of the driver's
// element arrays.  Same values as the reference's Arrays::init_data
// (cxx/pointers_only/data_structures.cpp:42-92 == fortran/main.F90:103-154); written
// field-by-field over flat index ranges instead of the reference's nested AT_ macros.
#include "homme_data.hpp"

#include <cmath>
#include <cstddef>
#include <vector>

namespace Homme {

int num_elems = 10;

namespace {
constexpr std::size_t PP = std::size_t(np) * np;
constexpr std::size_t BLK = PP * nlev;

real* zeros(std::size_t n) { return new real[n](); }
}  // namespace

void Arrays::init_data() {
  const std::size_t ne = num_elems;
  elem_D = zeros(ne * PP * 4);
  elem_Dinv = zeros(ne * PP * 4);
  elem_fcor = zeros(ne * PP);
  elem_spheremp = zeros(ne * PP);
  elem_metdet = zeros(ne * PP);
  elem_rmetdet = zeros(ne * PP);
  elem_state_dp3d = zeros(ne * timelevels * BLK);
  elem_state_v = zeros(ne * timelevels * BLK * 2);
  elem_state_T = zeros(ne * timelevels * BLK);
  elem_state_phis = zeros(ne * PP);
  elem_state_Qdp = zeros(ne * qsize_d * 2 * BLK);
  elem_derived_eta_dot_dpdn = zeros(ne * (BLK + PP));
  elem_derived_omega_p = zeros(ne * BLK);
  elem_derived_phi = zeros(ne * BLK);
  elem_derived_pecnd = zeros(ne * BLK);
  elem_derived_vn0 = zeros(ne * BLK * 2);

  // per-point tables shared by every element (1-based indices as doubles)
  std::vector<real> fcor(PP), phi0(PP), gi(PP), gj(PP);
  for (int i = 0; i < np; ++i)
    for (int j = 0; j < np; ++j) {
      const real ii = i + 1, jj = j + 1;
      gi[i * np + j] = ii;
      gj[i * np + j] = jj;
      fcor[i * np + j] = std::sin(ii + jj);
      phi0[i * np + j] = std::cos(ii + 3 * jj);
    }

  for (std::size_t e = 0; e < ne; ++e) {
    const real ee = real(e + 1);
    for (std::size_t q = 0; q < PP; ++q) {
      const real ii = gi[q], jj = gj[q];
      const std::size_t eq = e * PP + q;
      elem_fcor[eq] = fcor[q];
      elem_metdet[eq] = ii * jj;
      elem_rmetdet[eq] = 1. / elem_metdet[eq];
      elem_spheremp[eq] = 2 * ii;
      elem_state_phis[eq] = ii + jj;
      elem_D[eq * 4 + 0] = 1.0;
      elem_D[eq * 4 + 3] = 2.0;
      elem_Dinv[eq * 4 + 0] = 1.0;
      elem_Dinv[eq * 4 + 3] = 0.5;
    }
    for (int l = 0; l < nlev; ++l) {
      const real ll = l + 1;
      for (std::size_t q = 0; q < PP; ++q) {
        const real ii = gi[q], jj = gj[q];
        const std::size_t o = e * BLK + std::size_t(l) * PP + q;
        elem_derived_phi[o] = phi0[q] + ll;
        elem_derived_vn0[2 * o] = 1.0;
        elem_derived_vn0[2 * o + 1] = 1.0;
        elem_derived_pecnd[o] = 1.0;
        elem_derived_omega_p[o] = jj * jj;
        elem_state_Qdp[e * qsize_d * 2 * BLK + std::size_t(l) * PP + q] = 1.0 + std::sin(ii * jj * ll);
        for (int t = 0; t < timelevels; ++t) {
          const real tt = t + 1;
          const std::size_t s = (e * timelevels + t) * BLK + std::size_t(l) * PP + q;
          elem_state_dp3d[s] = 10.0 * ll + ee + ii + jj + tt;
          elem_state_v[2 * s] = 1.0 + 0.5 * ll + ii + jj + 0.2 * ee + 2.0 * tt;
          elem_state_v[2 * s + 1] = 1.0 + 0.5 * ll + ii + jj + 0.2 * ee + 3.0 * tt;
          elem_state_T[s] = 1000.0 - ll - ii - jj + 0.1 * ee + tt;
        }
      }
    }
  }
}

void release_host_mapping();  // homme_caar.cpp

void Arrays::cleanup_data() {
  release_host_mapping();  // page locks compute_and_apply_rhs(TestData&) may hold on these arrays
  real** all[] = {&elem_D, &elem_Dinv, &elem_fcor, &elem_spheremp, &elem_metdet, &elem_rmetdet,
                  &elem_state_dp3d, &elem_state_v, &elem_state_T, &elem_state_phis, &elem_state_Qdp,
                  &elem_derived_eta_dot_dpdn, &elem_derived_omega_p, &elem_derived_phi,
                  &elem_derived_pecnd, &elem_derived_vn0};
  for (real** p : all) {
    delete[] *p;
    *p = nullptr;
  }
}

void Constants::init_data() {
  Rwater_vapor = 461.5;
  Rgas = 287.04;
  cp = 1005.0;
  kappa = Rgas / cp;
  rrearth = 1.0 / 6.376e6;
  eta_ave_w = 1.0;
}

void Control::init_data() {
  nets = 0;
  nete = num_elems;
  n0 = 0;
  np1 = 1;
  nm1 = 2;
  qn0 = 0;
  dt2 = 1.0;
}

void HVCoord::init_data() {
  ps0 = 10.0;
  for (int i = 0; i < nlevp; ++i) hyai[i] = nlev + 1 - i;
}

// Gauss-Lobatto-Legendre derivative matrix, Dvv[i][j] = l_j'(x_i).  The reference
// tabulates it for np=4 only (data_structures.cpp:152-162: values[j*np+i], indexed out of
// bounds for any other np); for np=4 the literals are used so the driver reproduces the
// reference's printed norms digit for digit.
void Derivative::init_data() {
  if (np == 4) {
    static const real lit[16] = {
        -3.0000000000000000, -0.80901699437494745, 0.30901699437494745, -0.50000000000000000,
        4.0450849718747373,  0.00000000000000000,  -1.11803398874989490, 1.54508497187473700,
        -1.5450849718747370, 1.11803398874989490,  0.00000000000000000,  -4.04508497187473730,
        0.5000000000000000,  -0.30901699437494745, 0.80901699437494745,  3.000000000000000000};
    for (int i = 0; i < np; ++i)
      for (int j = 0; j < np; ++j) Dvv[i][j] = lit[(j * 4 + i) % 16];
    return;
  }
  const int N = np - 1;
  std::vector<real> x(np), L(np);
  auto legendre = [](int n, real t) {
    real p0 = 1.0, p1 = t;
    if (n == 0) return p0;
    for (int k = 2; k <= n; ++k) {
      const real pk = ((2.0 * k - 1.0) * t * p1 - (k - 1.0) * p0) / k;
      p0 = p1;
      p1 = pk;
    }
    return p1;
  };
  const real pi = 3.14159265358979323846;
  for (int i = 0; i <= N; ++i) {
    real t = -std::cos(pi * i / N);
    if (i > 0 && i < N)
      for (int it = 0; it < 100; ++it) {  // Newton on (1-t^2) P_N'(t) = N (P_{N-1} - t P_N)
        const real f = N * (legendre(N - 1, t) - t * legendre(N, t));
        const real fp = -real(N) * (N + 1) * legendre(N, t);
        const real dt = f / fp;
        t -= dt;
        if (std::fabs(dt) < 1e-16) break;
      }
    x[i] = (i == 0) ? -1.0 : (i == N ? 1.0 : t);
    L[i] = legendre(N, x[i]);
  }
  for (int i = 0; i <= N; ++i)
    for (int j = 0; j <= N; ++j) {
      if (i != j) Dvv[i][j] = L[i] / (L[j] * (x[i] - x[j]));
      else if (i == 0) Dvv[i][j] = -0.25 * N * (N + 1);
      else if (i == N) Dvv[i][j] = 0.25 * N * (N + 1);
      else Dvv[i][j] = 0.0;
    }
}

void TestData::init_data() {
  arrays.init_data();
  constants.init_data();
  control.init_data();
  hvcoord.init_data();
  deriv.init_data();
}

void TestData::update_time_levels() {
  const int old_np1 = control.np1;
  control.np1 = control.nm1;
  control.nm1 = control.n0;
  control.n0 = old_np1;
}

void TestData::cleanup_data() { arrays.cleanup_data(); }

}  // namespace Homme
