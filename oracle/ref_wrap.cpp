// ref_wrap.cpp — C entry points around the REFERENCE's own serial C++ path.
//
// TEST INFRASTRUCTURE ONLY.  oracle/Makefile compiles this file together with
// the reference sources compute_and_apply_rhs.cpp, sphere_operators.cpp and
// data_structures.cpp *where they lie* under
// /root/reference/compute_and_apply_rhs_test/cxx/pointers_only/ into
// oracle/_ref/libref_caar_np<NP>_nlev<PLEV>.so.  No reference source is copied
// into this repository; this file only includes the reference's headers and
// calls its functions, so the reference can be driven on caller-owned arrays
// (fixtures for tests/golden, validation of oracle/caar_oracle.c, and the
// "reference" CPU baseline of bench.py).
//
// The reference fixes NP/PLEV at compile time (dimensions.hpp:9-15 via
// config.h), hence one library per (NP, PLEV).
#include "compute_and_apply_rhs.hpp"
#include "data_structures.hpp"
#include "sphere_operators.hpp"

#include <cstring>

namespace Homme {
// Defined in the reference's main.cpp:11 (not compiled here); read by
// Arrays::init_data and Control::init_data (data_structures.cpp:12).
int num_elems = 0;
}  // namespace Homme

namespace {

// Order = member order of Homme::Arrays (data_structures.hpp:18-44).
void bind_arrays(Homme::Arrays& a, double* const p[16]) {
  a.elem_D = p[0];
  a.elem_Dinv = p[1];
  a.elem_fcor = p[2];
  a.elem_spheremp = p[3];
  a.elem_metdet = p[4];
  a.elem_rmetdet = p[5];
  a.elem_state_dp3d = p[6];
  a.elem_state_v = p[7];
  a.elem_state_T = p[8];
  a.elem_state_phis = p[9];
  a.elem_state_Qdp = p[10];
  a.elem_derived_eta_dot_dpdn = p[11];
  a.elem_derived_omega_p = p[12];
  a.elem_derived_phi = p[13];
  a.elem_derived_pecnd = p[14];
  a.elem_derived_vn0 = p[15];
}

void array_lengths(int ne, long len[16]) {
  using namespace Homme;
  const long pp = np * np, blk = (long)nlev * pp;
  len[0] = len[1] = ne * pp * 4;
  len[2] = len[3] = len[4] = len[5] = ne * pp;
  len[6] = (long)ne * timelevels * blk;
  len[7] = (long)ne * timelevels * blk * 2;
  len[8] = (long)ne * timelevels * blk;
  len[9] = ne * pp;
  len[10] = (long)ne * qsize_d * 2 * blk;
  len[11] = (long)ne * nlevp * pp;
  len[12] = len[13] = len[14] = (long)ne * blk;
  len[15] = (long)ne * blk * 2;
}

void fill_scalars(Homme::TestData& d, int nets, int nete, int n0, int np1, int nm1, int qn0,
                  double dt2, double rrearth, double eta_ave_w, double Rwater_vapor, double Rgas,
                  double kappa, double ps0, const double* hyai, const double* Dvv) {
  using namespace Homme;
  d.control.nets = nets;
  d.control.nete = nete;
  d.control.n0 = n0;
  d.control.np1 = np1;
  d.control.nm1 = nm1;
  d.control.qn0 = qn0;
  d.control.dt2 = dt2;
  d.constants.rrearth = rrearth;
  d.constants.eta_ave_w = eta_ave_w;
  d.constants.Rwater_vapor = Rwater_vapor;
  d.constants.Rgas = Rgas;
  d.constants.cp = Rgas / kappa;
  d.constants.kappa = kappa;
  d.hvcoord.ps0 = ps0;
  for (int i = 0; i < nlevp; ++i) d.hvcoord.hyai[i] = hyai[i];
  std::memcpy(&d.deriv.Dvv[0][0], Dvv, sizeof(double) * np * np);
}

}  // namespace

extern "C" {

void ref_dims(int out[4]) {
  out[0] = Homme::np;
  out[1] = Homme::nlev;
  out[2] = Homme::qsize_d;
  out[3] = Homme::timelevels;
}

// Runs the reference's TestData::init_data() (data_structures.cpp:165-172) for
// num_elems elements and copies every array, the scalars and Dvv out.
// scalars[12] = rrearth, eta_ave_w, cp, Rwater_vapor, Rgas, kappa, dt2, ps0,
//               n0, np1, nm1, qn0 (the four ints as doubles)
void ref_init_data(int num_elems, double* const arrays[16], double* scalars, double* hyai,
                   double* Dvv) {
  using namespace Homme;
  Homme::num_elems = num_elems;
  TestData d;
  d.arrays.init_data();
  d.constants.init_data();
  d.control.init_data();
  d.hvcoord.init_data();
  if (np == 4) d.deriv.init_data();  // the literals are np=4 only (data_structures.cpp:152-162)
  long len[16];
  array_lengths(num_elems, len);
  double* const src[16] = {d.arrays.elem_D, d.arrays.elem_Dinv, d.arrays.elem_fcor,
                           d.arrays.elem_spheremp, d.arrays.elem_metdet, d.arrays.elem_rmetdet,
                           d.arrays.elem_state_dp3d, d.arrays.elem_state_v, d.arrays.elem_state_T,
                           d.arrays.elem_state_phis, d.arrays.elem_state_Qdp,
                           d.arrays.elem_derived_eta_dot_dpdn, d.arrays.elem_derived_omega_p,
                           d.arrays.elem_derived_phi, d.arrays.elem_derived_pecnd,
                           d.arrays.elem_derived_vn0};
  for (int i = 0; i < 16; ++i) std::memcpy(arrays[i], src[i], sizeof(double) * len[i]);
  scalars[0] = d.constants.rrearth;
  scalars[1] = d.constants.eta_ave_w;
  scalars[2] = d.constants.cp;
  scalars[3] = d.constants.Rwater_vapor;
  scalars[4] = d.constants.Rgas;
  scalars[5] = d.constants.kappa;
  scalars[6] = d.control.dt2;
  scalars[7] = d.hvcoord.ps0;
  scalars[8] = d.control.n0;
  scalars[9] = d.control.np1;
  scalars[10] = d.control.nm1;
  scalars[11] = d.control.qn0;
  for (int i = 0; i < nlevp; ++i) hyai[i] = d.hvcoord.hyai[i];
  if (np == 4) std::memcpy(Dvv, &d.deriv.Dvv[0][0], sizeof(double) * np * np);
  d.cleanup_data();
}

// Homme::compute_and_apply_rhs (compute_and_apply_rhs.cpp:15) on caller-owned arrays.
void ref_compute_and_apply_rhs(double* const arrays[16], int nets, int nete, int n0, int np1,
                               int nm1, int qn0, double dt2, double rrearth, double eta_ave_w,
                               double Rwater_vapor, double Rgas, double kappa, double ps0,
                               const double* hyai, const double* Dvv) {
  Homme::TestData d;
  bind_arrays(d.arrays, arrays);
  fill_scalars(d, nets, nete, n0, np1, nm1, qn0, dt2, rrearth, eta_ave_w, Rwater_vapor, Rgas,
               kappa, ps0, hyai, Dvv);
  Homme::compute_and_apply_rhs(d);
}

// The reference driver's loop (main.cpp:113-121: `reps` back-to-back calls on the same TestData, the rotation commented
// out) in one C call: bench.py's threaded CPU baseline runs it once per host thread, so no Python sits between the calls.
void ref_compute_and_apply_rhs_repeat(double* const arrays[16], int nets, int nete, int n0, int np1, int nm1, int qn0,
                                      double dt2, double rrearth, double eta_ave_w, double Rwater_vapor, double Rgas,
                                      double kappa, double ps0, const double* hyai, const double* Dvv, int reps) {
  Homme::TestData d;
  bind_arrays(d.arrays, arrays);
  fill_scalars(d, nets, nete, n0, np1, nm1, qn0, dt2, rrearth, eta_ave_w, Rwater_vapor, Rgas, kappa, ps0, hyai, Dvv);
  for (int i = 0; i < reps; ++i) Homme::compute_and_apply_rhs(d);
}

// The three sphere operators (sphere_operators.cpp:9,50,91) for element `ie` of
// caller-owned geometry arrays.  which: 0 gradient (in np*np, out np*np*2),
// 1 divergence (in np*np*2, out np*np), 2 vorticity (in np*np*2, out np*np).
void ref_sphere_operator(int which, const double* in, double* out, double* const arrays[16],
                         int ie, double rrearth, const double* Dvv) {
  Homme::TestData d;
  bind_arrays(d.arrays, arrays);
  d.constants.rrearth = rrearth;
  std::memcpy(&d.deriv.Dvv[0][0], Dvv, sizeof(double) * Homme::np * Homme::np);
  if (which == 0) Homme::gradient_sphere(in, d, ie, out);
  if (which == 1) Homme::divergence_sphere(in, d, ie, out);
  if (which == 2) Homme::vorticity_sphere(in, d, ie, out);
}

// print_results_2norm's arithmetic (compute_and_apply_rhs.cpp:372-399) is not
// callable without the print; compute_norm (…:353) is.
double ref_compute_norm(const double* field, int length) {
  return Homme::compute_norm(field, length);
}

}  // extern "C"
