// homme_data.hpp — host-side data model of the CAAR driver, layout-compatible with the
// reference's Homme::TestData (compute_and_apply_rhs_test/cxx/pointers_only/
// data_structures.hpp:10-89) so that code written against the reference's structs —
// including the reference's own main.cpp — works unchanged on top of the HIP library.
//
// Dimensions are compile-time constants as in the reference (dimensions.hpp:9-15, from
// config.h); override with -DCAAR_NP=.. -DCAAR_PLEV=.. when building for another
// configuration (supported on the GPU: NP=4 with PLEV 72 or 128, NP=8 with PLEV 72).
#ifndef HOMME_DATA_HPP
#define HOMME_DATA_HPP

#ifndef CAAR_NP
#define CAAR_NP 4
#endif
#ifndef CAAR_PLEV
#define CAAR_PLEV 72
#endif
#ifndef CAAR_QSIZE_D
#define CAAR_QSIZE_D 1
#endif
#ifndef CAAR_NUM_TIME_LEVELS
#define CAAR_NUM_TIME_LEVELS 3
#endif

namespace Homme {

typedef double real;

constexpr int np = CAAR_NP;
constexpr int qsize_d = CAAR_QSIZE_D;
constexpr int nlev = CAAR_PLEV;
constexpr int nlevp = nlev + 1;
constexpr int timelevels = CAAR_NUM_TIME_LEVELS;

extern int num_elems;  // set by the driver before init_data (reference: main.cpp:11)

struct HVCoord {
  real ps0;
  real hyai[nlevp];
  void init_data();
};

// 16 element-major arrays; member order is part of the interface (include/caar.h CaarArrays)
struct Arrays {
  real* elem_D;
  real* elem_Dinv;
  real* elem_fcor;
  real* elem_spheremp;
  real* elem_metdet;
  real* elem_rmetdet;

  real* elem_state_dp3d;
  real* elem_state_v;
  real* elem_state_T;
  real* elem_state_phis;
  real* elem_state_Qdp;

  real* elem_derived_eta_dot_dpdn;
  real* elem_derived_omega_p;
  real* elem_derived_phi;
  real* elem_derived_pecnd;
  real* elem_derived_vn0;

  void init_data();
  void cleanup_data();
};

struct Constants {
  real rrearth;
  real eta_ave_w;
  real cp;
  real Rwater_vapor;
  real Rgas;
  real kappa;
  void init_data();
};

struct Control {
  int nets;
  int nete;
  int n0;
  int np1;
  int nm1;
  int qn0;
  real dt2;
  void init_data();
};

struct Derivative {
  real Dvv[np][np];
  void init_data();
};

struct TestData {
  Arrays arrays = {};
  Constants constants = {};
  Control control = {};
  Derivative deriv = {};
  HVCoord hvcoord = {};

  void init_data();
  void update_time_levels();
  void cleanup_data();
};

}  // namespace Homme
#endif
