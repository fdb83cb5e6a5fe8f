/*
 * caar.h — C ABI of the MI355X (gfx950) compute_and_apply_rhs implementation: THE BOUNDARY.
 *
 * FROZEN at CAAR_ABI_VERSION 6 (round 5): what a host binds to.  Nothing is removed from or changed in this header
 * without bumping the version; tuning, placement and measurement entry points live in caar_tuning.h and are NOT part of
 * the frozen surface (tests/test_host.py pins the exported set against INTEGRATION.md section 3).
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

#define CAAR_ABI_VERSION 6

enum {
  CAAR_OK = 0,
  CAAR_EINVAL = -1,       /* null pointer, bad index or range */
  CAAR_EUNSUPPORTED = -2, /* (np, nlev) — or its rsplit == 0 form — has no compiled kernel: see caar_supported[_ex]() */
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

/* 1 if a kernel exists for (np, nlev), else 0.  np=4: kernels specialised for nlev 72 and 128 and one with a run-time
 * level count for every other nlev in 2..256 (the reference builds any PLEV, config.h.in:3; libcaar_hip_extra.so, the
 * -DCAAR_EXTRA_NLEV=1 build, adds shapes specialised for nlev 26, 30, 32, 60, 64, 80, 96); np=8: nlev 72. */
int caar_supported(int np, int nlev);
/* The same for one vertical form: rsplit as CaarParams::rsplit.  rsplit > 0 (vertically Lagrangian, the reference's
 * path): == caar_supported.  rsplit == 0 (Eulerian): the default library serves np=4 up to 128 levels and np=8; beyond
 * 128 levels its form spills registers and exists only in libcaar_hip_extra.so.  Every launch entry point refuses an
 * unsupported combination with CAAR_EUNSUPPORTED before anything is enqueued (also inside caar_run_steps: no capture is
 * started).  (ABI 6) */
int caar_supported_ex(int np, int nlev, int rsplit);
/* CAAR_ABI_VERSION the library was built with. */
int caar_abi_version(void);
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
 * one kernel launch where the selected variant has a step-loop kernel and nsteps >= 2 (default build: NP=4 NLEV 72 / 128,
 * NP=8 NLEV 72; caar_tuning.h caar_set_fused_steps), else
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

/* print_results_2norm's per-element arithmetic (P:353-390) on device-resident arrays:
 * out_dev[3*(e-e0)+f] = pow(compute_norm(field_f of element e at time level tl), 2),
 * f = 0,1,2 for v, T, dp3d.  `out_dev` is a DEVICE buffer of 3*(e1-e0) doubles.
 * Asynchronous on `stream`; the caller sums over elements and takes the root (P:394-396). */
int caar_launch_state_norms(const CaarDims *dims, const CaarArrays *dev, int tl, int e0, int e1,
                            double *out_dev, void *stream);

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

/* ---- device arrays for hosts that allocate themselves ------------------------------------------
 * Allocates the 16 element arrays for `dims` on HIP device `device` and returns their DEVICE pointers in *out_dev, placed
 * the way the path runs fastest (where the arrays lie in HBM is worth 3-5 % on MI355X, DESIGN.md section 5 "Placement";
 * caar_create allocates the same way).  How they are placed is a tuning matter: CaarPlacement / caar_arrays_alloc_ex in
 * caar_tuning.h.  caar_arrays_free unmaps and releases everything; the calling thread's current device is left as it
 * was; it returns non-zero if a HIP call of the teardown failed (the memory is then leaked rather than reused). */
typedef struct CaarArena CaarArena;
int caar_arrays_alloc(CaarArena **arena, const CaarDims *dims, int device, CaarArrays *out_dev);
int caar_arrays_free(CaarArena *arena);

/* ---- context API: the library owns the device copies ---------------------------
 * What Homme::compute_and_apply_rhs(TestData&) needs when TestData lives in host
 * memory (Arrays::init_data, data_structures.cpp:14-31). */
typedef struct CaarContext CaarContext;

/* Allocates device storage for dims->num_elems elements on HIP device `device`
 * and a private stream. */
int caar_create(CaarContext **ctx, const CaarDims *dims, int device);
/* Releases the context.  Address-space policy (also caar_arrays_free): a placed arena's virtual address range stays
 * RESERVED for the life of the process — every chunk is unmapped and its physical memory released (all return codes
 * checked), but the range is never handed back, so no address is ever mapped twice.  A conservative policy, not a
 * demonstrated driver defect: round 2 saw wrong results after an arena had been freed and a new one mapped; the bare-HIP
 * probe with a control arm (tools/probes/vmm_va_reuse_probe.hip, profiles/r04/vmm_va_reuse_probe.log) reads correct data
 * through a re-used range in every arm, so the cause is unexplained.  Cost: the arrays' size rounded up to 64 MiB per array
 * (2 GiB for a 10 000-element NP=4 NLEV=72 set) of the 128 TiB address space per destroyed arena; no memory. */
void caar_destroy(CaarContext *ctx);
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
/* Wait for everything enqueued on the context stream. */
int caar_sync(CaarContext *ctx);
/* Device pointers / stream of the context (for callers that launch their own work). */
int caar_device_arrays(CaarContext *ctx, CaarArrays *out);
void *caar_stream(CaarContext *ctx);
/* print_results_2norm's arithmetic on the device (P:372-399): out[0..2] =
 * ||v||_2, ||T||_2, ||dp3d||_2 of time level `tl` over elements [e0, e1), Kahan sum
 * of squares per element as compute_norm (P:353-370).  Synchronous. */
int caar_state_norms(CaarContext *ctx, int tl, int e0, int e1, double out[3]);
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
