/*
 * caar_oracle.h — CPU oracle for compute_and_apply_rhs (CAAR).
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * serial algorithm, used as the parity checker by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg.  Nothing in the
 * product path (tinman_sandbox_amd/, include/) may include, link or call it.
 *
 * Parity status: PINNED for the vertically-Lagrangian path (rsplit > 0: everything the
 * reference builds); "parity unpinned" for the rsplit == 0 extension (see oracle_params).
 * The restatement is checked (tests/test_oracle.py)
 *   - against the reference's own golden vectors Ttest/v1test/v2test
 *     (compute_and_apply_rhs_test/fortran/test_mod.F90:8-882), and
 *   - against outputs of the reference's C++ (cxx/pointers_only) and Fortran
 *     (fortran/routine_mod.F90) implementations compiled from where they lie
 *     by oracle/Makefile into oracle/_ref/, committed as tests/golden/ fixtures.
 *
 * All arrays use the reference's C++ element-major layout
 * (cxx/pointers_only/data_structures.hpp:18-44, data_structures.cpp:14-31):
 *   elem_D, elem_Dinv                 [ie][np][np][2][2]
 *   fcor, spheremp, metdet, rmetdet   [ie][np][np]
 *   state_dp3d, state_T               [ie][timelevels][nlev][np][np]
 *   state_v                           [ie][timelevels][nlev][np][np][2]
 *   state_phis                        [ie][np][np]
 *   state_Qdp                         [ie][qsize_d][2][nlev][np][np]
 *   derived_eta_dot_dpdn              [ie][nlev+1][np][np]
 *   derived_omega_p, phi, pecnd       [ie][nlev][np][np]
 *   derived_vn0                       [ie][nlev][np][np][2]
 * Unlike the reference (compile-time NP/PLEV from config.h) every dimension
 * is a run-time argument so one library serves all BASELINE.json configs.
 */
#ifndef CAAR_ORACLE_H
#define CAAR_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Same member order as Homme::Arrays (data_structures.hpp:18-44). */
typedef struct oracle_arrays {
  double *elem_D;
  double *elem_Dinv;
  double *elem_fcor;
  double *elem_spheremp;
  double *elem_metdet;
  double *elem_rmetdet;
  double *elem_state_dp3d;
  double *elem_state_v;
  double *elem_state_T;
  double *elem_state_phis;
  double *elem_state_Qdp;
  double *elem_derived_eta_dot_dpdn;
  double *elem_derived_omega_p;
  double *elem_derived_phi;
  double *elem_derived_pecnd;
  double *elem_derived_vn0;
} oracle_arrays;

/* Homme::Control + Constants + HVCoord + Derivative (data_structures.hpp:10-76)
 * flattened; Dvv is row-major Dvv[i][j], np*np doubles. */
typedef struct oracle_params {
  int np, nlev, qsize_d, timelevels;
  int nets, nete;          /* element range [nets, nete) */
  int n0, np1, nm1, qn0;   /* 0-based time-level indices; qn0 == -1 -> dry */
  double dt2;
  double rrearth, eta_ave_w, Rwater_vapor, Rgas, kappa;
  double ps0, hyai0;       /* only hyai[0] enters the path (P:84) */
  const double *Dvv;
  /* rsplit > 0: vertically Lagrangian, the only branch the reference's built variants have
   * (P:22-28: eta_dot_dpdn, T_vadv, v_vadv = 0).  rsplit == 0: Eulerian vertical coordinate,
   * stated in the reference only in routine_extracted.F90:224-262,515-517 and
   * level_vectorized_ppscan/CaarFunctor.hpp:505-547, neither of which it builds or tests:
   * PARITY UNPINNED for that branch.  hybi: nlev+1 interface coefficients (hybvcoord_mod.F90:19),
   * read only when rsplit == 0. */
  int rsplit;
  const double *hybi;
} oracle_params;

/* sphere_operators.cpp:9-48 / derivative_mod_base.F90:25-65 */
void oracle_gradient_sphere(int np, const double *s, const double *Dvv,
                            const double *Dinv, double rrearth, double *ds);
/* sphere_operators.cpp:50-89 / derivative_mod_base.F90:182-230 */
void oracle_divergence_sphere(int np, const double *v, const double *Dvv,
                              const double *Dinv, const double *metdet,
                              const double *rmetdet, double rrearth, double *div);
/* sphere_operators.cpp:91-129 / derivative_mod_base.F90:127-177 */
void oracle_vorticity_sphere(int np, const double *v, const double *Dvv,
                             const double *D, const double *rmetdet,
                             double rrearth, double *vort);
/* compute_and_apply_rhs.cpp:280-312 / routine_mod.F90:255-293 */
void oracle_preq_hydrostatic(int np, int nlev, const double *phis,
                             const double *T_v, const double *p,
                             const double *dp, double Rgas, double *phi);
/* compute_and_apply_rhs.cpp:314-352 / routine_mod.F90:207-252 */
void oracle_preq_omega_ps(int np, int nlev, const double *p,
                          const double *vgrad_p, const double *divdp,
                          double *omega_p);
/* compute_and_apply_rhs.cpp:15-278 / routine_mod.F90:7-193.
 * Returns 0, or -1 if scratch allocation failed. */
int oracle_compute_and_apply_rhs(const oracle_arrays *a, const oracle_params *p);
/* `reps` back-to-back calls (the reference driver's loop, main.cpp:113-121); stops at the first non-zero return. */
int oracle_compute_and_apply_rhs_repeat(const oracle_arrays *a, const oracle_params *p, int reps);

/* Kahan 2-norm, compute_and_apply_rhs.cpp:353-370 / utils_mod.F90:9-30 */
double oracle_compute_norm(const double *field, long length);
/* print_results_2norm without the print, compute_and_apply_rhs.cpp:372-399:
 * out[0..2] = ||v||, ||T||, ||dp|| of time level np1 over [nets,nete). */
void oracle_state_norms(const oracle_arrays *a, const oracle_params *p, double out[3]);

/* Closed-form synthetic initialiser, data_structures.cpp:42-92 (== main.F90:103-154). */
void oracle_init_arrays(const oracle_arrays *a, int np, int nlev, int qsize_d,
                        int timelevels, int num_elems);
/* NP=4 derivative matrix literals, data_structures.cpp:152-162.
 * f32_rounded != 0 reproduces the Fortran driver (main.F90:83-96), whose
 * literals are default-real (float32) widened to double. */
void oracle_init_dvv_np4(double *Dvv, int f32_rounded);
/* Gauss-Lobatto-Legendre derivative matrix for any np (own construction: the
 * reference hard-codes np=4 only).  Dvv[i][j] = l'_j(x_i), which reproduces
 * the np=4 literals above to rounding. */
void oracle_init_dvv_gll(int np, double *Dvv);

/* ---- sphere operators next to the CAAR path, sphere_ops_oracle.c.  PARITY UNPINNED
 * (see that file's header).  One np x np level per call; arrays in this repository's layout:
 * scalar [np][np], vector [np][np][2], tensors [np][np][2][2], vec_sph2cart [np][np][3][2].
 * K: = cxx/level_vectorized_ppscan/SphereOperators.hpp. */
void oracle_k_gradient_sphere(int np, const double *s, const double *Dvv, const double *Dinv, double rrearth,
                              double *grad);                                                   /* K:229-269 */
void oracle_gradient_sphere_update(int np, const double *s, const double *Dvv, const double *Dinv, double rrearth,
                                   double *grad);                                              /* K:271-312 */
void oracle_k_divergence_sphere(int np, const double *v, const double *Dvv, const double *Dinv, const double *metdet,
                                double rrearth, double *div);                                  /* K:315-358 */
void oracle_divergence_sphere_update(int np, double alpha, double beta, const double *v, const double *Dvv,
                                     const double *Dinv, const double *metdet, double rrearth, double *div); /* K:363-403 */
void oracle_k_vorticity_sphere_vector(int np, const double *v, const double *Dvv, const double *D, const double *metdet,
                                      double rrearth, double *vort);                           /* K:452-490 */
void oracle_divergence_sphere_wk(int np, const double *v, const double *Dvv, const double *Dinv, const double *spheremp,
                                 double rrearth, double *div);                                 /* K:494-534 */
void oracle_laplace_simple(int np, const double *s, const double *Dvv, const double *Dinv, const double *spheremp,
                           double rrearth, double *lap);                                       /* K:538-550 */
void oracle_laplace_tensor(int np, const double *s, const double *Dvv, const double *Dinv, const double *spheremp,
                           const double *tensorVisc, double rrearth, double *lap);             /* K:556-596 */
void oracle_curl_sphere_wk_testcov(int np, const double *s, const double *Dvv, const double *D, const double *mp,
                                   double rrearth, double *curls);                             /* K:640-690 */
void oracle_grad_sphere_wk_testcov(int np, const double *s, const double *Dvv, const double *D, const double *mp,
                                   const double *metinv, const double *metdet, double rrearth, double *grads); /* K:694-770 */
void oracle_vlaplace_sphere_wk_cartesian(int np, const double *v, const double *Dvv, const double *Dinv,
                                         const double *spheremp, const double *tensorVisc, const double *vec_sph2cart,
                                         double rrearth, int undamp_rr, double *lap);          /* K:849-915 (K:777-844) */
void oracle_vlaplace_sphere_wk_contra(int np, const double *v, const double *Dvv, const double *D, const double *Dinv,
                                      const double *mp, const double *spheremp, const double *metinv, const double *metdet,
                                      double nu_ratio, double rrearth, double *lap);           /* K:938-993 */

/* EulerStepFunctor.hpp:32-68, one element: qtens = Qdp(qn0) - dt * div(vstar * Qdp(qn0)) per tracer and level */
void oracle_euler_step(int np, int nlev, int qsize, int qn0, double dt, const double *vstar, const double *qdp,
                       const double *Dvv, const double *Dinv, const double *metdet, double rrearth, double *qtens);
void oracle_laplace_tensor_replace(int np, const double *Dvv, const double *Dinv, const double *spheremp,
                                   const double *tensorVisc, double rrearth, double *laplace);

#ifdef __cplusplus
}
#endif
#endif
