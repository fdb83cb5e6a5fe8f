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

// hybrid vertical coordinate: only ps0 and hyai(1) enter the path (data_structures.hpp:10-16)
struct HVCoord {
  real ps0, hyai[nlevp];
  void init_data();
};

// The 16 element-major arrays.  Member ORDER is part of the interface: include/caar.h's CaarArrays is read
// as this struct (and as the reference's, data_structures.hpp:18-44).
struct Arrays {
  real *elem_D, *elem_Dinv;                                          // [ie][np][np][2][2]
  real *elem_fcor, *elem_spheremp, *elem_metdet, *elem_rmetdet;      // [ie][np][np]
  real *elem_state_dp3d;                                             // [ie][timelevels][nlev][np][np]
  real *elem_state_v;                                                // [ie][timelevels][nlev][np][np][2]
  real *elem_state_T;                                                // as dp3d
  real *elem_state_phis;                                             // [ie][np][np]
  real *elem_state_Qdp;                                              // [ie][qsize_d][2][nlev][np][np]
  real *elem_derived_eta_dot_dpdn;                                   // [ie][nlevp][np][np]
  real *elem_derived_omega_p, *elem_derived_phi, *elem_derived_pecnd;  // [ie][nlev][np][np]
  real *elem_derived_vn0;                                            // [ie][nlev][np][np][2]

  void init_data();     // new[] + the reference's closed-form values (data_structures.cpp:14-92)
  void cleanup_data();  // drops the shim's page locks, then frees
};

struct Constants {  // data_structures.hpp:46-56
  real rrearth, eta_ave_w, cp, Rwater_vapor, Rgas, kappa;
  void init_data();
};

struct Control {  // data_structures.hpp:58-69
  int nets, nete;     // element range [nets, nete)
  int n0, np1, nm1;   // time levels
  int qn0;            // tracer time level, -1 = dry
  real dt2;
  void init_data();
};

struct Derivative {  // data_structures.hpp:71-76
  real Dvv[np][np];
  void init_data();
};

struct TestData {  // data_structures.hpp:78-89
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
