/*
 * caar.h — C ABI of the MI355X (gfx950) compute_and_apply_rhs implementation.
 *
 * libcaar_hip.so is the drop-in boundary for the reference's hot path
 *     namespace Homme { void compute_and_apply_rhs(TestData& data); }
 *     (compute_and_apply_rhs_test/cxx/pointers_only/compute_and_apply_rhs.hpp:9,
 *      Fortran twin fortran/routine_mod.F90:7)
 * The reference has no FFI layer of its own; these entry points are
 * what a binding for that one function needs: plain pointers, ints and doubles,
 * no C++ or torch types, no exceptions.  Every function returns 0 on success, a
 * positive hipError_t value if the HIP runtime failed, or a negative CAAR_E* code.
 *
 * Data layout is the reference's element-major C++ layout, unchanged
 * (data_structures.hpp:18-44, data_structures.cpp:14-31), so a host that already
 * owns a Homme::Arrays can pass its 16 pointers as they are:
 *     elem_D, elem_Dinv                          [ie][np][np][2][2]
 *     elem_fcor, spheremp, metdet, rmetdet       [ie][np][np]
 *     elem_state_dp3d, elem_state_T              [ie][timelevels][nlev][np][np]
 *     elem_state_v                               [ie][timelevels][nlev][np][np][2]
 *     elem_state_phis                            [ie][np][np]
 *     elem_state_Qdp                             [ie][qsize_d][2][nlev][np][np]
 *     elem_derived_eta_dot_dpdn                  [ie][nlev+1][np][np]
 *     elem_derived_omega_p, phi, pecnd           [ie][nlev][np][np]
 *     elem_derived_vn0                           [ie][nlev][np][np][2]
 */
#ifndef CAAR_H
#define CAAR_H

#ifdef __cplusplus
extern "C" {
#endif

#define CAAR_ABI_VERSION 5

enum {
  CAAR_OK = 0,
  CAAR_EINVAL = -1,       /* null pointer, bad index or range */
  CAAR_EUNSUPPORTED = -2, /* (np, nlev) has no compiled kernel: see caar_supported() */
  CAAR_ENODEVICE = -3,    /* no HIP device / wrong architecture */
  CAAR_ENOMEM = -4
};

/* The 16 element arrays.  Member order == Homme::Arrays
 * (data_structures.hpp:18-44).  Pointers are host or device pointers depending on
 * the call (stated per function). */
typedef struct CaarArrays {
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
} CaarArrays;
#define CAAR_NUM_ARRAYS 16

/* Compile-time dimensions of the reference (dimensions.hpp:9-15 via config.h),
 * run-time here. */
typedef struct CaarDims {
  int np;         /* GLL points per element edge: 4 or 8            */
  int nlev;       /* vertical levels (PLEV)                          */
  int qsize_d;    /* tracer slots in elem_state_Qdp (>= 1)           */
  int timelevels; /* time levels in elem_state_* (reference: 3)      */
  int num_elems;  /* elements the arrays hold                         */
} CaarDims;

/* Homme::Control (data_structures.hpp:58-69) + Constants (:46-56) + the parts of
 * HVCoord (:10-16) and Derivative (:71-76) the path reads.  All host values. */
typedef struct CaarParams {
  int nets, nete;      /* element range [nets, nete), 0-based (Control::nets/nete)   */
  int n0, np1, nm1;    /* 0-based time-level indices                                 */
  int qn0;             /* Qdp time slot (0/1), or -1 for the dry branch (P:128)      */
  double dt2;
  double rrearth;      /* Constants::rrearth                                         */
  double eta_ave_w;
  double Rwater_vapor;
  double Rgas;
  double kappa;
  double ps0;          /* HVCoord::ps0                                               */
  double hyai0;        /* HVCoord::hyai[0], the only hyai entry the path reads (P:84) */
  const double *Dvv;   /* HOST pointer, np*np doubles, row-major Dvv[i][j]           */
  /* Vertical coordinate (Control::rsplit, level_vectorized_ppscan/Control.hpp:48-49).
   * rsplit > 0: vertically Lagrangian — eta_dot_dpdn, T_vadv, v_vadv are zero; this is the
   * only branch the reference's built variants contain (P:22-28) and what every golden
   * vector covers.  rsplit == 0: Eulerian — the interface mass flux and the vertical
   * advection of T and v are computed (fortran/routine_extracted.F90:224-262,515-517;
   * preq_vertadv, level_vectorized_ppscan/CaarFunctor.hpp:505-547); the reference states
   * that branch only in files it does not build, so its parity is UNPINNED.
   * Must be set: a zero-initialised struct asks for rsplit == 0 and is refused
   * (CAAR_EINVAL) unless the hybi pointer the call needs is given.  NP=4 and NP=8. */
  int rsplit;
  const double *hybi;     /* HOST pointer, nlev+1 interface coefficients (hybvcoord_mod.F90:19);
                           * read when rsplit == 0 by caar_run / caar_run_mapped             */
  const double *hybi_dev; /* DEVICE pointer to the same values; read when rsplit == 0 by the
                           * stateless caar_launch (as dvv_dev: the caller uploads it once)  */
} CaarParams;

/* 1 if a kernel exists for (np, nlev), else 0.  np=4: kernels specialised for nlev 26, 30, 32,
 * 60, 64, 72, 80, 96, 128 and one with a run-time level count for every other nlev in 2..256
 * (the reference builds any PLEV, config.h.in:3); np=8: nlev 72. */
int caar_supported(int np, int nlev);
/* CAAR_ABI_VERSION the library was built with. */
int caar_abi_version(void);
/* Debug builds (libcaar_hip_debug.so, compiled with -DCAAR_DEBUG: `python -m tinman_sandbox_amd.build --debug`): the
 * reference's only hot-path assertion, check_dp3d (level_vectorized_ppscan/CaarFunctor.hpp:82-97: dp3d(np1) > 0 under
 * !NDEBUG).  The kernels count every dp3d(np1) they store that is not positive (a counter, not a trap: a trapping kernel
 * takes the GPU down); this returns the count on the current device since the start / the last reset, after waiting for
 * the device.  -1 in a release build, -2 if the HIP calls fail. */
long long caar_debug_dp3d_violations(int reset);
/* Number of HIP devices visible to the process (0 if none / no driver). */
int caar_device_count(void);
/* Static text for a return code. */
const char *caar_strerror(int rc);
/* Number of doubles in array `index` (0..15, CaarArrays member order) for `dims`,
 * or -1. */
long long caar_array_len(const CaarDims *dims, int index);
/* Algorithmic HBM bytes one element update moves:
 * 8*(21*np^2*nlev + 2*np^2*(nlev+1) + 13*np^2); the dry branch reads one block less. */
long long caar_algorithmic_bytes(int np, int nlev, int dry);

/* ---- stateless launch on device-resident arrays -------------------------------
 * Replaces one call of Homme::compute_and_apply_rhs (P:15) for elements
 * [params->nets, params->nete).  `dev` holds DEVICE pointers to arrays in the
 * layout above (e.g. torch tensors or hipMalloc'd buffers); `stream` is a
 * hipStream_t (NULL = default stream).  Asynchronous: returns after enqueueing.
 * `dvv_dev` is a DEVICE buffer of np*np doubles holding params->Dvv (the caller
 * uploads it once; params->Dvv is ignored here).  elem_state_v and elem_derived_vn0 must
 * be 16-byte aligned (they are moved as (u, v) pairs), the rest 8-byte.  No allocation, no
 * synchronisation, safe to capture in a hipGraph. */
int caar_launch(const CaarDims *dims, const CaarArrays *dev, const double *dvv_dev,
                const CaarParams *params, void *stream);

/* `nsteps` consecutive calls (the driver loop main.cpp:113-121), with TestData::update_time_levels
 * (data_structures.cpp:174-180: np1, nm1, n0 <- nm1, n0, np1) between them if rotate != 0, on device-resident arrays:
 * one kernel launch where the selected variant has a step-loop kernel and nsteps >= 2 (see caar_set_fused_steps), else
 * nsteps launches of caar_launch (also for rsplit == 0 and for a non-finite eta_ave_w).  Same arguments and rules as caar_launch;
 * bit-identical to nsteps calls of it, in every array. */
int caar_launch_steps(const CaarDims *dims, const CaarArrays *dev, const double *dvv_dev, const CaarParams *params,
                      int nsteps, int rotate, void *stream);

/* The three sphere operators on their own (reference: sphere_operators.hpp:9-16,
 * gradient_sphere / divergence_sphere / vorticity_sphere(field, data, ielem, out)), batched
 * over `nlevels` fields of element `ie`.  which = 0 gradient: in [lev][np][np] -> out
 * [lev][np][np][2]; 1 divergence, 2 vorticity: in [lev][np][np][2] -> out [lev][np][np].
 * `dev` supplies elem_D, elem_Dinv, elem_metdet, elem_rmetdet (device); in/out are device
 * buffers.  Runs the same device functions the fused kernel uses.  Asynchronous. */
int caar_sphere_operator(const CaarDims *dims, const CaarArrays *dev, const double *dvv_dev, int which,
                         int ie, int nlevels, const double *in_dev, double *out_dev, double rrearth,
                         void *stream);

/* The same for the elements [e0, e1) in one launch: in [e][lev][np][np](,2) -> out
 * [e][lev][np][np](,2), e = 0 .. e1-e0-1, `nlevels` fields per element (bandwidth-bound:
 * every input byte read once, every output byte written once). */
int caar_sphere_operator_range(const CaarDims *dims, const CaarArrays *dev, const double *dvv_dev, int which,
                               int e0, int e1, int nlevels, const double *in_dev, double *out_dev,
                               double rrearth, void *stream);

/* ---- the sphere operators next to the CAAR path ---------------------------------
 * Reference: cxx/level_vectorized_ppscan/SphereOperators.hpp:271-993 (Kokkos device functions the reference
 * defines but never builds, calls or tests: their parity is UNPINNED, see oracle/sphere_ops_oracle.c).
 * Same batching as caar_sphere_operator_range: elements [e0, e1), `nlevels` fields per element,
 * in/out [e - e0][lev][np][np] (scalar) or [e - e0][lev][np][np][2] (vector), this repository's index
 * convention (field[a][b] == Fortran (a+1, b+1), as the pointers_only arrays).
 *   code                                   in -> out        geometry read                       reference
 *   0  GRADIENT_SPHERE                     s  -> v          Dinv                                 K:229-269
 *   1  DIVERGENCE_SPHERE                   v  -> s          Dinv, metdet, rmetdet                K:315-358
 *   2  VORTICITY_SPHERE (vector input)     v  -> s          D, rmetdet                           K:452-490
 *   3  DIVERGENCE_SPHERE_WK                v  -> s          Dinv, spheremp                       K:494-534
 *   4  LAPLACE_SIMPLE                      s  -> s          Dinv, spheremp                       K:538-550
 *   5  LAPLACE_TENSOR                      s  -> s          Dinv, spheremp, tensorVisc           K:556-596
 *   6  CURL_SPHERE_WK_TESTCOV              s  -> v          D, mp                                K:640-690
 *   7  GRAD_SPHERE_WK_TESTCOV              s  -> v          D, mp, metinv, metdet                K:694-770
 *   8  VLAPLACE_SPHERE_WK_CONTRA           v  -> v          D, Dinv, mp, spheremp, metinv, metdet, rmetdet; nu_ratio   K:938-993
 *   9  VLAPLACE_SPHERE_WK_CARTESIAN        v  -> v          Dinv, spheremp, tensorVisc, vec_sph2cart   K:849-915 (rigid-rotation term kept, K:891)
 *   10 GRADIENT_SPHERE_UPDATE              s  -> v += grad  Dinv                                 K:271-312
 *   11 DIVERGENCE_SPHERE_UPDATE            v  -> s = beta*s + alpha*div   Dinv, metdet, rmetdet  K:363-403
 *   12 VLAPLACE_SPHERE_WK_CARTESIAN_DAMPED v  -> v          as 9, without the rigid-rotation term  K:777-844
 *   13 LAPLACE_TENSOR_REPLACE              s  -> s in place  as 5; out_dev is input AND output, in_dev is ignored   K:600-637
 * Codes 1, 2, 8, 11 multiply by rmetdet (the pointers_only operators' form, sphere_operators.cpp:85,125)
 * where K: forms 1/metdet on the fly.  The EulerStep functor (EulerStepFunctor.hpp:32-68): caar_euler_step below. */
enum {
  CAAR_OP_GRADIENT_SPHERE = 0,
  CAAR_OP_DIVERGENCE_SPHERE = 1,
  CAAR_OP_VORTICITY_SPHERE = 2,
  CAAR_OP_DIVERGENCE_SPHERE_WK = 3,
  CAAR_OP_LAPLACE_SIMPLE = 4,
  CAAR_OP_LAPLACE_TENSOR = 5,
  CAAR_OP_CURL_SPHERE_WK_TESTCOV = 6,
  CAAR_OP_GRAD_SPHERE_WK_TESTCOV = 7,
  CAAR_OP_VLAPLACE_SPHERE_WK_CONTRA = 8,
  CAAR_OP_VLAPLACE_SPHERE_WK_CARTESIAN = 9,
  CAAR_OP_GRADIENT_SPHERE_UPDATE = 10,
  CAAR_OP_DIVERGENCE_SPHERE_UPDATE = 11,
  CAAR_OP_VLAPLACE_SPHERE_WK_CARTESIAN_DAMPED = 12,
  CAAR_OP_LAPLACE_TENSOR_REPLACE = 13,
  CAAR_OP_COUNT = 14
};
/* Per-element geometry, DEVICE pointers to element 0, point-major with the components fastest like
 * CaarArrays: D, Dinv, metinv, tensorVisc [ie][np][np][2][2]; metdet, rmetdet, spheremp, mp [ie][np][np];
 * vec_sph2cart [ie][np][np][3][2].  Only the arrays the chosen operator reads (table above) must be set;
 * the 2x2 / 3x2 ones must be 16-byte aligned. */
typedef struct CaarOperatorGeometry {
  const double *D, *Dinv, *metdet, *rmetdet, *spheremp, *mp, *metinv, *tensorVisc, *vec_sph2cart;
} CaarOperatorGeometry;
typedef struct CaarOperatorScalars {
  double rrearth;
  double alpha, beta; /* DIVERGENCE_SPHERE_UPDATE */
  double nu_ratio;    /* VLAPLACE_SPHERE_WK_CONTRA */
} CaarOperatorScalars;
/* Asynchronous on `stream`.  in_dev / out_dev 16-byte aligned, not overlapping (code 13 works in place on out_dev; in_dev may
 * be NULL). */
int caar_sphere_operator_ex(const CaarDims *dims, const CaarOperatorGeometry *geo, const double *dvv_dev, int which,
                            int e0, int e1, int nlevels, const double *in_dev, double *out_dev,
                            const CaarOperatorScalars *scalars, void *stream);

/* The tracer step sketched in EulerStepFunctor.hpp:32-68 (Kokkos variant; not compilable as it stands — its call
 * of divergence_sphere_update, E:65-66, passes 8 arguments to the 9-parameter K:363-370 — so this follows what the
 * functor STATES, parity unpinned): for every tracer q < qsize and every level
 *     vstar_qdp = vstar * Qdp(qn0, q)   (E:59-60)      qtens(q) = Qdp(qn0, q)   (E:61)
 *     qtens(q)  = 1.0 * qtens(q) + (-dt) * divergence_sphere(vstar_qdp)          (E:65-66, K:398-399)
 * fused into one pass: vstar is read once per level for all tracers, no vstar_qdp buffer.
 * vstar_dev [e1-e0][nlev][np][np][2] (16-byte aligned), Qdp_dev = CaarArrays.state_Qdp of element 0
 * ([ie][qsize_d][2][nlev][np][np], dims->qsize_d), qtens_dev [e1-e0][qsize][nlev][np][np]; geo: Dinv, metdet,
 * rmetdet of element 0.  Asynchronous on `stream`. */
int caar_euler_step(const CaarDims *dims, const CaarOperatorGeometry *geo, const double *dvv_dev, int e0, int e1,
                    int qsize, int qn0, double dt, double rrearth, const double *vstar_dev, const double *Qdp_dev,
                    double *qtens_dev, void *stream);

/* The two vertical integrals of the path as functions of their own (reference:
 * compute_and_apply_rhs.hpp:11-17, P:280-352), batched over `nelem` columns-of-elements:
 * phis [e][np][np]; T_v, p, dp, phi, vgrad_p, divdp, omega_p [e][nlev][np][np] (device).  One thread per
 * column in the reference's own order, no FMA contraction, IEEE division: bit-identical to the reference. */
int caar_preq_hydrostatic(const CaarDims *dims, int nelem, const double *phis_dev, const double *T_v_dev,
                          const double *p_dev, const double *dp_dev, double Rgas, double *phi_dev, void *stream);
int caar_preq_omega_ps(const CaarDims *dims, int nelem, const double *p_dev, const double *vgrad_p_dev,
                       const double *divdp_dev, double *omega_p_dev, void *stream);

/* The reference's operator functions with their own calling convention — HOST pointers, one np x np field of one
 * element (one element's columns) in, one out, synchronous (sphere_operators.hpp:9-16: gradient_sphere /
 * divergence_sphere / vorticity_sphere(field, data, ielem, out); compute_and_apply_rhs.hpp:11-17: preq_hydrostatic,
 * preq_omega_ps).  `host` supplies elem_D, elem_Dinv, elem_metdet, elem_rmetdet as HOST arrays, dvv_host np*np
 * doubles; which = 0/1/2 as caar_sphere_operator.  Each call uploads a few hundred bytes to a library-owned
 * scratch area on device 0, launches the device operator and downloads the result: latency-bound by
 * construction, thread-safe (mutex).  What Homme::gradient_sphere(...) etc. of the C++ shim call. */
int caar_sphere_operator_host(const CaarDims *dims, const CaarArrays *host, const double *dvv_host, int which, int ie,
                              const double *in_host, double *out_host, double rrearth);
int caar_preq_hydrostatic_host(const CaarDims *dims, const double *phis, const double *T_v, const double *p,
                               const double *dp, double Rgas, double *phi);
int caar_preq_omega_ps_host(const CaarDims *dims, const double *p, const double *vgrad_p, const double *divdp,
                            double *omega_p);

/* Numerics hook: out[i] = the kernels' reciprocal of in[i] (v_rcp_f64 + two Newton steps,
 * used for the divisions by p and dp3d, P:150,219,291,323; <= 1 ulp for normal inputs). */
int caar_reciprocal(const double *in_dev, double *out_dev, long long n, void *stream);

/* print_results_2norm's per-element arithmetic (P:353-390) on device-resident arrays:
 * out_dev[3*(e-e0)+f] = pow(compute_norm(field_f of element e at time level tl), 2),
 * f = 0,1,2 for v, T, dp3d.  `out_dev` is a DEVICE buffer of 3*(e1-e0) doubles.
 * Asynchronous on `stream`; the caller sums over elements and takes the root (P:394-396). */
int caar_launch_state_norms(const CaarDims *dims, const CaarArrays *dev, int tl, int e0, int e1,
                            double *out_dev, void *stream);

/* Name of the kernel caar_launch dispatches for (np, nlev) (for profiles), or NULL. */
const char *caar_kernel_name(int np, int nlev);
/* Tuning: each (np, nlev) is compiled in a few launch shapes (tiles per wavefront,
 * register budget => workgroups per CU).  Variant 0 is the default; all variants
 * compute the same thing.  Process-wide; the selection is an atomic that every launch reads once, so
 * selecting while other threads launch is safe (a launch uses the old or the new variant). */
int caar_num_variants(int np, int nlev);
int caar_select_variant(int np, int nlev, int variant);
int caar_selected_variant(int np, int nlev);
const char *caar_variant_info(int np, int nlev, int variant);
/* Workgroup -> element mapping: 0 deals consecutive elements round-robin over the XCDs (all XCDs sweep the arrays
 * together); 1 gives each XCD one contiguous eighth of the element range; -1 (default) what the selected variant was
 * measured faster with (the default kernels of NP=4 NLEV=72 / 128 and NP=8: 1, +0.7..1.3 %; all other variants: 0).  Same results either way. */
int caar_set_xcd_chunked(int on);
/* Hybrid cache policy of the default NP=4 kernels: all element data streams with non-temporal
 * loads and stores, which do not allocate in the 256 MB memory-side Infinity Cache — except the
 * three read-modify-write accumulators (derived_vn0, omega_p, eta_dot_dpdn) of `bytes` worth of
 * elements, which use the default policy: a host that calls again
 * on the same arrays finds them in the cache instead of in HBM (one read and one write saved per
 * byte and call).  Default 224 MiB (best of a sweep: flat from 192 to 240 MiB); 0 makes every access streaming.  Same
 * results either way.  Process-wide, atomic, read once per launch (as the variant selection).
 * The window is a budget of the DEVICE, not of a launch (ABI 5): which elements are kept is a property of the element's
 * index in the arrays (an evenly spread subset of dims->num_elems), so launches on sub-ranges [nets, nete) of one array
 * set — HOMME's horizontal OpenMP threads (data_structures.hpp:58-69), one after the other or side by side on several
 * streams — together keep what one launch over everything keeps.  Contexts (caar_create) on one device share the window
 * in proportion to their sizes (caar_context_cache_window: this context's part).  A host that passes several SEPARATE
 * array sets to the stateless caar_launch on one device divides the budget itself (caar_set_cache_window(total / sets)). */
#define CAAR_CACHE_WINDOW_DEFAULT (224LL << 20)
int caar_set_cache_window(long long bytes);
long long caar_get_cache_window(void);
/* Adaptive window (default on; ABI 5).  Whether the window pays depends on what the host runs BETWEEN two calls: kept
 * accumulators are worth +13 % when the next call finds them (the routine replayed, or alternated with kernels that
 * stream), and -2 % when a neighbour with the default cache policy has evicted them.  The library therefore measures:
 * per array set (keyed on elem_derived_vn0), whole-range launches of a hybrid-policy kernel are now and then bracketed
 * by HIP events that are polled, never waited for; after 48 calls, whenever the current policy's kernel time drifts up
 * by more than 3 %, every 96 calls while the policy is all-streaming and every 4 096 while it is the window, the other policy runs for 7 calls and the
 * current one again for 7, and the faster becomes the policy (the window on ties).  Launches inside a stream capture,
 * on a sub-range, or through caar_run_steps' captured graph use the set's current policy and measure nothing; array sets
 * whose traffic per call fits the 256 MB cache whole (up to ~1 250 elements at NP=4 NLEV=72) are not tuned at all.  Same
 * results either way (both policies are the same kernel).  caar_set_adaptive_window(0): the window always applies.
 * caar_adaptive_window_state: the policy in force for the array set whose derived_vn0 is `vn0_dev` (1 window, 0 all
 * streaming; -1 if the set is unknown) and, where the pointers are not NULL, the medians of the last probe (ms; 0 before
 * the first) and the number of probes decided so far.  caar_adaptive_window_reset forgets every array set (a host that
 * changes its call pattern need not call it: drift and re-probes follow; benchmarks that switch patterns use it). */
int caar_set_adaptive_window(int on);
int caar_get_adaptive_window(void);
int caar_adaptive_window_state(const double *vn0_dev, double *ms_window, double *ms_streaming, long long *probes);
int caar_adaptive_window_reset(void);

/* ---- Fortran-layout ingest / egress -----------------------------------------------
 * A Fortran host holds the same 16 arrays with the FIRST index fastest
 * (fortran/element_state_mod.F90:17-23, element_mod.F90:69-121), flattened over elements:
 *     v(np,np,2,nlev,timelevels,ne)  T,dp3d(np,np,nlev,timelevels,ne)  Qdp(np,np,nlev,qsize_d,2,ne)
 *     phi,omega_p,pecnd(np,np,nlev,ne)  vn0(np,np,2,nlev,ne)  eta_dot_dpdn(np,np,nlev+1,ne)
 *     D,Dinv(np,np,2,2,ne)  fcor,spheremp,metdet,rmetdet,phis(np,np,ne)
 * (cf. the pull/push functions of level_vectorized_ppscan/Elements.cpp:154-435).
 * `f90_dev` holds DEVICE pointers to such arrays, in the CaarArrays member order;
 * these calls convert elements [e0, e1) to / from the C++ layout used by caar_launch.
 * caar_layout_to_f90 with all_arrays == 0 converts only the seven arrays the path
 * mutates.  Asynchronous on `stream`; source and destination must not overlap. */
int caar_layout_from_f90(const CaarDims *dims, const CaarArrays *f90_dev, const CaarArrays *caar_dev,
                         int e0, int e1, void *stream);
int caar_layout_to_f90(const CaarDims *dims, const CaarArrays *caar_dev, const CaarArrays *f90_dev,
                       int e0, int e1, int all_arrays, void *stream);

/* ---- measurement utilities (roofline context; never on the product path) ---------
 * caar_stream_copy: device copy of n_doubles with 8 or 16 bytes per lane — the measured
 * HBM ceiling next to the spec peak and the calibration run for the HBM PMC counters.
 * caar_traffic_skeleton: touches exactly the bytes caar_launch touches (NP=4; NP=8 NLEV=72), same
 * addressing and access widths, no arithmetic; it OVERWRITES the output arrays with
 * meaningless values.  `variant` picks the launch shape / cache policy being probed
 * (0 = the shape of the default kernel); unknown variants return hipErrorInvalidValue. */
int caar_stream_copy(double *dst_dev, const double *src_dev, long long n_doubles, int lane_bytes,
                     void *stream);
/* Tuned device copy of n_doubles (even; 16-byte aligned buffers): 16 bytes per lane, several
 * independent loads in flight per lane, contiguous 1 KiB wave segments, grid = CUs x resident
 * workgroups; `variant` in [0, caar_stream_copy_tuned_variants()) picks unroll / cache policy /
 * grid size (caar_stream_copy_tuned_info: text).  The ceiling bench.py quotes next to the spec
 * peak is the best of these on the box it runs on. */
int caar_stream_copy_tuned(double *dst_dev, const double *src_dev, long long n_doubles, int variant,
                           void *stream);
int caar_stream_copy_tuned_variants(void);
const char *caar_stream_copy_tuned_info(int variant);
int caar_traffic_skeleton(const CaarDims *dims, const CaarArrays *dev, const CaarParams *params,
                          int variant, void *stream);

/* ---- device arrays placed for bandwidth ----------------------------------------------------
 * Allocates the 16 element arrays for `dims` on HIP device `device` and returns their DEVICE pointers in *out_dev.
 * Where the arrays lie in HBM matters on MI355X: device memory falls into a few large address classes and the path runs
 * 3-5 % faster when its traffic is split over several of them than when all arrays lie in one (DESIGN.md section 5
 * "Placement").  So the arrays are backed, through HIP virtual memory management, by 64 MiB physical chunks sampled
 * evenly from a temporary pool (bounded: see CaarPlacement below), every array contiguous in virtual memory and at
 * least 2 MiB-aligned.  Data sets below 256 MiB, policy CAAR_PLACE_MALLOC, or a failing VMM route fall back to one
 * hipMalloc per array.  caar_create allocates this way too (caar_create_ex takes the same CaarPlacement).
 * caar_arrays_placement: 1 if the arena is chunk-backed (and the pool size / chunk size it used), 0 if plain. */
typedef struct CaarArena CaarArena;
/* How an allocation is placed — a per-call choice (a NULL pointer, or policy CAAR_PLACE_DEFAULT, means: spread, unless
 * the environment says CAAR_PLACEMENT=malloc).
 *   pool_bytes          upper bound of the temporary pool (0: CAAR_PLACEMENT_POOL_GIB from the environment, else
 *                       CAAR_PLACEMENT_POOL_DEFAULT).  The pool exists only while the call runs; what the arena keeps
 *                       afterwards is the arrays' own size rounded up to 64 MiB per array.
 *   max_free_fraction   the pool never takes more than this share of the device memory that is FREE when the call is
 *                       made (0: one half; at most 0.9), so a second rank on the same GPU or another allocator in the
 *                       process keeps the rest.  If the device fills up anyway while the pool is created, the call
 *                       still succeeds with a narrower spread, or with plain allocations.
 * A placed arena is mapped for its own device only: other GPUs (peer access over xGMI, RCCL buffers) cannot address
 * it.  Hosts that need peer-visible arrays pass CAAR_PLACE_MALLOC. */
enum { CAAR_PLACE_DEFAULT = 0, CAAR_PLACE_SPREAD = 1, CAAR_PLACE_MALLOC = 2 };
#define CAAR_PLACEMENT_POOL_DEFAULT (128LL << 30)
typedef struct CaarPlacement {
  int policy;
  long long pool_bytes;
  double max_free_fraction;
} CaarPlacement;
int caar_arrays_alloc(CaarArena **arena, const CaarDims *dims, int device, CaarArrays *out_dev);
int caar_arrays_alloc_ex(CaarArena **arena, const CaarDims *dims, int device, const CaarPlacement *placement,
                         CaarArrays *out_dev);
/* Unmaps and releases everything; the calling thread's current device is left as it was.  Returns non-zero if a
 * HIP call of the teardown failed (the memory is then leaked rather than reused). */
int caar_arrays_free(CaarArena *arena);
int caar_arrays_placement(const CaarArena *arena, long long *pool_chunks, long long *chunk_bytes);

/* ---- context API: the library owns the device copies ---------------------------
 * What Homme::compute_and_apply_rhs(TestData&) needs when TestData lives in host
 * memory (Arrays::init_data, data_structures.cpp:14-31). */
typedef struct CaarContext CaarContext;

/* Allocates device storage for dims->num_elems elements on HIP device `device`
 * and a private stream. */
int caar_create(CaarContext **ctx, const CaarDims *dims, int device);
/* The same with the placement of the device arrays chosen by the caller (NULL: as caar_create). */
int caar_create_ex(CaarContext **ctx, const CaarDims *dims, int device, const CaarPlacement *placement);
/* Releases the context.  Address-space policy (also caar_arrays_free): a placed arena's virtual address range stays
 * RESERVED for the life of the process — every chunk is unmapped and its physical memory released (all return codes
 * checked), but the range is never handed back, so no address is ever mapped twice.  A conservative policy, not a
 * demonstrated driver defect: round 2 saw wrong results after an arena had been freed and a new one mapped; the bare-HIP
 * probe with a control arm (tools/probes/vmm_va_reuse_probe.hip, profiles/r04/vmm_va_reuse_probe.log) reads correct data
 * through a re-used range in every arm, so the cause is unexplained.  Cost: the arrays' size rounded up to 64 MiB per array
 * (2 GiB for a 10 000-element NP=4 NLEV=72 set) of the 128 TiB address space per destroyed arena; no memory. */
void caar_destroy(CaarContext *ctx);
/* This context's part of the device's cache window (caar_set_cache_window), in bytes: all of it while it is the only
 * context on its device, else in proportion to the contexts' sizes.  -1 for a NULL context. */
long long caar_context_cache_window(CaarContext *ctx);
/* Host -> device copy of all 16 arrays for elements [e0, e1) (element-major layout:
 * one contiguous range per array).  `host` = pointers to element 0 of host arrays
 * holding at least e1 elements. Asynchronous on the context stream. */
int caar_upload(CaarContext *ctx, const CaarArrays *host, int e0, int e1);
/* Device -> host copy of the arrays the path mutates (state_v/T/dp3d, eta_dot_dpdn,
 * omega_p, phi, vn0) for elements [e0, e1); all_arrays != 0 copies all 16. */
int caar_download(CaarContext *ctx, const CaarArrays *host, int e0, int e1, int all_arrays);
/* The same two copies for a Fortran host: `f90_host` points to HOST arrays in Fortran order
 * (see "Fortran-layout ingest / egress" above); the re-layout happens on the device. */
int caar_upload_f90(CaarContext *ctx, const CaarArrays *f90_host, int e0, int e1);
int caar_download_f90(CaarContext *ctx, const CaarArrays *f90_host, int e0, int e1, int all_arrays);
/* caar_upload_f90 for a subset of the arrays: bit i of array_mask = array i in CaarArrays member order (pointers of
 * the others may be NULL).  A host that calls the path on arrays it owns uploads the constant geometry once and the
 * fields it changed each call (host/fortran/routine_mod_hip.F90). */
int caar_upload_f90_arrays(CaarContext *ctx, const CaarArrays *f90_host, int e0, int e1, unsigned array_mask);
/* Enqueue one compute_and_apply_rhs on the context's device arrays
 * (params->Dvv is read from host memory and cached on the device). */
int caar_run(CaarContext *ctx, const CaarParams *params);
/* `nsteps` compute_and_apply_rhs calls as ONE hipGraph launch on the context stream — for
 * hosts that step a small number of elements many times, where the per-call launch cost
 * (not the kernel) sets the pace.  rotate != 0 applies TestData::update_time_levels
 * (data_structures.cpp:174-180: np1, nm1, n0 <- nm1, n0, np1) between consecutive calls,
 * starting from params' indices; the caller rotates its own Control nsteps-1 times
 * afterwards, as after nsteps-1 single calls.  The graph is captured on first use and kept while
 * params, nsteps and rotate stay the same.  Asynchronous. */
int caar_run_steps(CaarContext *ctx, const CaarParams *params, int nsteps, int rotate);
/* How caar_run_steps / caar_launch_steps issue the calls.  1 (default): as ONE kernel launch where the selected variant
 * has a step-loop kernel (NP=4 NLEV 72, 128, 80, 64, 60 and NP=8 NLEV 72; rsplit > 0, finite eta_ave_w) and nsteps >= 2 (a
 * single call is faster through the tuned single launch: hybrid cache window, XCD preference) — elements are independent and every lane only ever touches its own points,
 * so each workgroup makes all nsteps calls for its element back to back: launch fill/drain once per nsteps instead of
 * once per call, the element's arrays still in cache from the second call on; bit-identical to single launches.
 * 0: always a hipGraph of nsteps single launches (what every other configuration uses).  Process-wide, atomic. */
int caar_set_fused_steps(int on);
int caar_get_fused_steps(void);
/* 1 if tuning variant `variant` of (np, nlev) has a step-loop kernel (NP=4 NLEV 72 / 128 / 80 / 64 / 60: the
 * two-workgroup shapes; NP=8 NLEV 72: the MFMA forms), else 0. */
int caar_has_fused_steps(int np, int nlev, int variant);
/* Wait for everything enqueued on the context stream. */
int caar_sync(CaarContext *ctx);
/* Device pointers / stream of the context (for callers that launch their own work). */
int caar_device_arrays(CaarContext *ctx, CaarArrays *out);
void *caar_stream(CaarContext *ctx);
/* print_results_2norm's arithmetic on the device (P:372-399): out[0..2] =
 * ||v||_2, ||T||_2, ||dp3d||_2 of time level `tl` over elements [e0, e1), Kahan sum
 * of squares per element as compute_norm (P:353-370).  Synchronous. */
int caar_state_norms(CaarContext *ctx, int tl, int e0, int e1, double out[3]);
/* Time `reps` back-to-back caar_run calls with hipEvents on the context stream;
 * *ms_total receives the elapsed milliseconds.  Synchronous. */
int caar_time_runs(CaarContext *ctx, const CaarParams *params, int reps, float *ms_total);

/* ---- host-mapped API: run directly on the host's arrays, no device copies ---------
 * For a host that calls the path like the reference does — Homme::compute_and_apply_rhs(
 * TestData&) on arrays it owns in host memory (P:15, data_structures.cpp:14-31) — the
 * cheapest route over PCIe is not upload + kernel + download (every array, every time
 * level, both directions) but the kernel itself reading and writing host memory: each
 * input byte crosses the link once, each output byte once, reads and writes overlap on
 * the full-duplex link, and no HBM is allocated.
 * caar_map_host page-locks the 16 host arrays (num_elems elements each, C++ layout) for
 * HIP device `device`; the host keeps ownership and may read/write them between calls.
 * caar_run_mapped performs one compute_and_apply_rhs on them and returns when the results
 * are visible to the host (synchronous, like the reference's function).
 * caar_unmap_host releases the page locks; call it before freeing the arrays. */
typedef struct CaarHostMapping CaarHostMapping;
int caar_map_host(CaarHostMapping **map, const CaarDims *dims, const CaarArrays *host, int device);
int caar_run_mapped(CaarHostMapping *map, const CaarParams *params);
int caar_unmap_host(CaarHostMapping *map);

#ifdef __cplusplus
}
#endif
#endif /* CAAR_H */
