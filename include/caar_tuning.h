/*
 * caar_tuning.h — tuning, placement, measurement and debug entry points of libcaar_hip.so.
 *
 * NOT part of the frozen boundary (include/caar.h, ABI 6): nothing a host needs to call — the defaults are the measured
 * best — and free to change between rounds.  Used by this repository's benchmarks, tests and tools (bench.py,
 * the scripts under tools/, tinman_sandbox_amd/caar.py); the host bindings (host/homme_caar.cpp, host/fortran/caar_mod.F90) include
 * and bind caar.h only.
 */
#ifndef CAAR_TUNING_H
#define CAAR_TUNING_H

#include "caar.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- kernel variants, element mapping, cache policy ------------------------------------------------ */
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
 * The window is a budget of the DEVICE, not of a launch (since ABI 5): which elements are kept is a property of the element's
 * index in the arrays (an evenly spread subset of dims->num_elems), so launches on sub-ranges [nets, nete) of one array
 * set — HOMME's horizontal OpenMP threads (data_structures.hpp:58-69), one after the other or side by side on several
 * streams — together keep what one launch over everything keeps.  Contexts (caar_create) on one device share the window
 * in proportion to their sizes (caar_context_cache_window: this context's part).  A host that passes several SEPARATE
 * array sets to the stateless caar_launch on one device divides the budget itself (caar_set_cache_window(total / sets)). */
#define CAAR_CACHE_WINDOW_DEFAULT (224LL << 20)
int caar_set_cache_window(long long bytes);
long long caar_get_cache_window(void);
/* Adaptive window (default on; since ABI 5).  Whether the window pays depends on what the host runs BETWEEN two calls: kept
 * accumulators are worth +13 % when the next call finds them (the routine replayed, or alternated with kernels that
 * stream), and -2 % when a neighbour with the default cache policy has evicted them.  The library therefore measures:
 * per array set (keyed on elem_derived_vn0), whole-range launches of a hybrid-policy kernel are now and then bracketed
 * by HIP events that are polled, never waited for; after 48 calls, whenever the current policy's kernel time drifts up
 * by more than 3 %, every 96 calls while the policy is all-streaming and every 4 096 while it is the window, the other policy runs for 7 calls and the
 * current one again for 7, and the faster becomes the policy (the window on ties).  Launches inside a stream capture,
 * on a sub-range, or through caar_run_steps' captured graph use the set's current policy and measure nothing (a captured
 * graph is re-captured when the policy it baked in is no longer the set's); array sets whose traffic per call fits the
 * 256 MB cache whole (up to ~1 250 elements at NP=4 NLEV=72) are not tuned at all.  Same results either way: where a
 * variant names a streaming twin (NP=4 NLEV 72 and 128: variant 1, the all-streaming instantiation of the same launch
 * shape — POL = 1 instead of 2, its own preferred XCD mapping; rsplit == 0 launches are never adapted)
 * the all-streaming policy launches that twin, bit-identical by construction (every kernel of one NP performs the same
 * roundings) and tested at both level counts.  caar_set_adaptive_window(0): the window always applies.
 * caar_adaptive_window_state: the policy in force for the array set whose derived_vn0 is `vn0_dev` (1 window, 0 all
 * streaming; -1 if the set is unknown) and, where the pointers are not NULL, the medians of the last probe (ms; 0 before
 * the first) and the number of probes decided so far.  caar_adaptive_window_reset forgets every array set (a host that
 * changes its call pattern need not call it: drift and re-probes follow; benchmarks that switch patterns use it).  A set's
 * entry is also forgotten when its memory is released (caar_arrays_free, caar_destroy). */
int caar_set_adaptive_window(int on);
int caar_get_adaptive_window(void);
/* How often a launch has taken the tuner's mutex since the process started (tests/host_reentrancy.cpp: sub-range launches
 * never do, whole-range launches only when a sample or a probe step is due). */
long long caar_adaptive_window_lock_count(void);
int caar_adaptive_window_state(const double *vn0_dev, double *ms_window, double *ms_streaming, long long *probes);
int caar_adaptive_window_reset(void);


/* This context's part of the device's cache window (caar_set_cache_window), in bytes: all of it while it is the only
 * context on its device, else in proportion to the contexts' sizes.  -1 for a NULL context. */
long long caar_context_cache_window(CaarContext *ctx);

/* How caar_run_steps / caar_launch_steps issue the calls.  1 (default): as ONE kernel launch where the selected variant
 * has a step-loop kernel (NP=4 NLEV 72, 128 and NP=8 NLEV 72; NLEV 80, 64, 60 only in libcaar_hip_extra.so, the
 * -DCAAR_EXTRA_NLEV=1 build; rsplit > 0, finite eta_ave_w) and nsteps >= 2 (a
 * single call is faster through the tuned single launch: hybrid cache window, XCD preference) — elements are independent and every lane only ever touches its own points,
 * so each workgroup makes all nsteps calls for its element back to back: launch fill/drain once per nsteps instead of
 * once per call, the element's arrays still in cache from the second call on; bit-identical to single launches.
 * 0: always a hipGraph of nsteps single launches (what every other configuration uses).  Process-wide, atomic. */
int caar_set_fused_steps(int on);
int caar_get_fused_steps(void);
/* 1 if tuning variant `variant` of (np, nlev) has a step-loop kernel (NP=4 NLEV 72 / 128: the two-workgroup shapes; NLEV 80 / 64 / 60
 * in the extra build only; NP=8 NLEV 72: the MFMA forms), else 0. */
int caar_has_fused_steps(int np, int nlev, int variant);

/* ---- placement of the device arrays ----------------------------------------------------------------
 * Allocates the 16 element arrays for `dims` on HIP device `device` and returns their DEVICE pointers in *out_dev.
 * Where the arrays lie in HBM matters on MI355X: device memory falls into a few large address classes and the path runs
 * 3-5 % faster when its traffic is split over several of them than when all arrays lie in one (DESIGN.md section 5
 * "Placement").  So the arrays are backed, through HIP virtual memory management, by 64 MiB physical chunks sampled
 * evenly from a temporary pool (bounded: see CaarPlacement below), every array contiguous in virtual memory and at
 * least 2 MiB-aligned.  Data sets below 256 MiB, policy CAAR_PLACE_MALLOC, or a failing VMM route fall back to one
 * hipMalloc per array.  caar_create allocates this way too (caar_create_ex takes the same CaarPlacement).
 * caar_arrays_placement: 1 if the arena is chunk-backed (and the pool size / chunk size it used), 0 if plain. */
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
int caar_arrays_alloc_ex(CaarArena **arena, const CaarDims *dims, int device, const CaarPlacement *placement,
                         CaarArrays *out_dev);
int caar_arrays_placement(const CaarArena *arena, long long *pool_chunks, long long *chunk_bytes);

/* The same with the placement of the device arrays chosen by the caller (NULL: as caar_create). */
int caar_create_ex(CaarContext **ctx, const CaarDims *dims, int device, const CaarPlacement *placement);

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

/* Time `reps` back-to-back caar_run calls with hipEvents on the context stream;
 * *ms_total receives the elapsed milliseconds.  Synchronous. */
int caar_time_runs(CaarContext *ctx, const CaarParams *params, int reps, float *ms_total);


/* Numerics hook: out[i] = the kernels' reciprocal of in[i] (v_rcp_f64 + two Newton steps,
 * used for the divisions by p and dp3d, P:150,219,291,323; <= 1 ulp for normal inputs). */
int caar_reciprocal(const double *in_dev, double *out_dev, long long n, void *stream);


/* ---- debug builds ------------------------------------------------------------------------------------ */
/* Debug builds (libcaar_hip_debug.so, compiled with -DCAAR_DEBUG: `python -m tinman_sandbox_amd.build --debug`): the
 * reference's only hot-path assertion, check_dp3d (level_vectorized_ppscan/CaarFunctor.hpp:82-97: dp3d(np1) > 0 under
 * !NDEBUG).  The kernels count every dp3d(np1) they store that is not positive (a counter, not a trap: a trapping kernel
 * takes the GPU down); this returns the count on the current device since the start / the last reset, after waiting for
 * the device.  -1 in a release build, -2 if the HIP calls fail. */
long long caar_debug_dp3d_violations(int reset);

#ifdef __cplusplus
}
#endif
#endif /* CAAR_TUNING_H */
