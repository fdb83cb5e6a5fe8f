// caar_abi.hip — the extern "C" boundary declared in include/caar.h.
//
// Thin: argument validation, kernel selection by (np, nlev), device bookkeeping
// for the context API.  No CPU fallback: if no HIP device or no kernel for the
// requested dimensions exists the call fails with an error code.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/caar.h"
#include "../../include/caar_tuning.h"
#include "caar_kernel_args.h"
#include "caar_window_tuner.h"

namespace caar {
extern KernelVariant kNp4Nlev72[];
extern int kNp4Nlev72Count;
extern KernelVariant kNp4Nlev128[];
extern int kNp4Nlev128Count;
extern KernelVariant kNp8Nlev72[];
extern int kNp8Nlev72Count;
#if CAAR_EXTRA_NLEV
extern KernelVariant kNp4Nlev32[], kNp4Nlev60[], kNp4Nlev64[], kNp4Nlev80[], kNp4Nlev96[], kNp4Nlev26[], kNp4Nlev30[];
extern int kNp4Nlev32Count, kNp4Nlev60Count, kNp4Nlev64Count, kNp4Nlev80Count, kNp4Nlev96Count, kNp4Nlev26Count, kNp4Nlev30Count;
#endif
extern KernelVariant kNp4NlevAny[];
hipError_t launch_state_norms(const double* v, const double* T, const double* dp, int np, int nlev,
                              int timelevels, int tl, int e0, int e1, double* out3_per_elem,
                              hipStream_t stream);

hipError_t launch_stream_copy(double* dst, const double* src, size_t n_doubles, int lane_bytes,
                              hipStream_t stream);
hipError_t launch_stream_copy_tuned(double* dst, const double* src, size_t n_doubles, int variant, hipStream_t stream);
int stream_copy_tuned_variants();
const char* stream_copy_tuned_info(int v);
hipError_t launch_reciprocal(const double* in, double* out, size_t n, hipStream_t s);
hipError_t launch_sphere_operator_ex(int np, int which, const OpArgs& a, hipStream_t s);
unsigned sphere_operator_ex_needs(int which);
hipError_t launch_euler_step(int np, const EulerArgs& a, hipStream_t s);
hipError_t launch_preq_hydrostatic(int np, int nlev, int nelem, const double* phis, const double* Tv, const double* p,
                                   const double* dp, double Rgas, double* phi, hipStream_t s);
hipError_t launch_preq_omega_ps(int np, int nlev, int nelem, const double* p, const double* vgrad_p, const double* divdp,
                                double* omega_p, hipStream_t s);
hipError_t launch_layout(double* dst, const double* src, size_t n, int np, int nc, int nlev, int qd,
                         int qdp_outer, bool to_caar, hipStream_t s);
hipError_t launch_layout_all(int np, int count, double* const* dst, const double* const* src, const size_t* n, const int* nc,
                             const int* qdp_outer, int nlev, int qd, bool to_caar, hipStream_t s);
hipError_t launch_traffic_skeleton(const KernelArgs& k, int nlev, int variant, int num_elems, hipStream_t stream);
hipError_t launch_traffic_skeleton_np8(const KernelArgs& k, int nlev, int variant, int num_elems, hipStream_t stream);
#ifdef CAAR_DEBUG
long long debug_dp3d_count_np4(int reset);
long long debug_dp3d_count_np4_steps(int reset);
long long debug_dp3d_count_np8(int reset);
#endif

// `selected` is the only mutable member: atomic, read ONCE per launch (launch_choice), so a
// caar_select_variant racing with launches on other threads is well defined (each launch uses
// either the old or the new variant, never a mixture).
struct Config {
  int np, nlev;
  const KernelVariant* variants;
  int count;
  std::atomic<int> selected;
};
static Config* configs(int* n) {
  // built once, thread-safe (C++11 static initialisation); the counts are link-time constants of
  // the kernel files
  static Config c[] = {
      {4, 72, kNp4Nlev72, kNp4Nlev72Count, {0}},
      {4, 128, kNp4Nlev128, kNp4Nlev128Count, {0}},
      {8, 72, kNp8Nlev72, kNp8Nlev72Count, {0}},
#if CAAR_EXTRA_NLEV
      {4, 32, kNp4Nlev32, kNp4Nlev32Count, {0}},
      {4, 60, kNp4Nlev60, kNp4Nlev60Count, {0}},
      {4, 64, kNp4Nlev64, kNp4Nlev64Count, {0}},
      {4, 80, kNp4Nlev80, kNp4Nlev80Count, {0}},
      {4, 96, kNp4Nlev96, kNp4Nlev96Count, {0}},
      {4, 26, kNp4Nlev26, kNp4Nlev26Count, {0}},
      {4, 30, kNp4Nlev30, kNp4Nlev30Count, {0}},
#endif
  };
  *n = (int)(sizeof(c) / sizeof(c[0]));
  return c;
}
static Config* find_config(int np, int nlev) {
  int n;
  Config* c = configs(&n);
  for (int i = 0; i < n; ++i)
    if (c[i].np == np && c[i].nlev == nlev) return &c[i];
  // NP=4: any other level count runs the kernel compiled for a run-time count
  static Config any = {4, 0, kNp4NlevAny, 1, {0}};
  if (np == 4 && nlev >= 2 && nlev <= 256) return &any;
  return nullptr;
}
}  // namespace caar

namespace caar {
// Device copies of the two small host arrays a call reads (Dvv; hybi when rsplit == 0),
// re-uploaded only when the host values change.
struct HostConstants {
  double* dev;  // [64] Dvv, then [nlev + 1] hybi
  std::vector<double>* host;
  bool dvv_valid, hybi_valid;

  hipError_t create(int nlev) {
    host = new (std::nothrow) std::vector<double>(64 + nlev + 1, 0.0);
    if (!host) return hipErrorOutOfMemory;
    return hipMalloc((void**)&dev, sizeof(double) * host->size());
  }
  void destroy() {
    if (dev) (void)hipFree(dev);
    delete host;
    dev = nullptr;
    host = nullptr;
  }
  // Fills the device-side fields of *p (a copy of the caller's params) and returns dvv_dev.
  // The staging vector lives as long as the owner, so the async copies' sources outlive the call.
  hipError_t sync(const CaarDims& d, CaarParams* p, hipStream_t stream, const double** dvv_dev) {
    const size_t n = (size_t)d.np * d.np;
    if (!dvv_valid || std::memcmp(host->data(), p->Dvv, sizeof(double) * n) != 0) {
      std::memcpy(host->data(), p->Dvv, sizeof(double) * n);
      hipError_t e = hipMemcpyAsync(dev, host->data(), sizeof(double) * n, hipMemcpyHostToDevice, stream);
      if (e != hipSuccess) return e;
      dvv_valid = true;
    }
    if (p->rsplit == 0) {
      const size_t m = (size_t)d.nlev + 1;
      if (!hybi_valid || std::memcmp(host->data() + 64, p->hybi, sizeof(double) * m) != 0) {
        std::memcpy(host->data() + 64, p->hybi, sizeof(double) * m);
        hipError_t e = hipMemcpyAsync(dev + 64, host->data() + 64, sizeof(double) * m, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return e;
        hybi_valid = true;
      }
      p->hybi_dev = dev + 64;
    }
    *dvv_dev = dev;
    return hipSuccess;
  }
};
}  // namespace caar

// Process-wide tuning knobs.  Atomics, read once per launch into a LaunchChoice, so setters may race
// with launches on other threads (HOMME's horizontal OpenMP calls the routine from several host
// threads): every launch sees one consistent set of values.
static std::atomic<int> g_xcd_chunked{-1};  // workgroup -> element mapping, see element_of_block(); -1: what the variant prefers
// bytes of element data the hybrid cache policy keeps in the memory-side cache (0: none, all streaming);
// default = best of a 0..320 MB sweep at NLEV 72 and 128 (the cache holds 256 MB, shared with everything else)
static std::atomic<long long> g_cache_window{CAAR_CACHE_WINDOW_DEFAULT};
// caar_run_steps: 1 = one fused launch where the selected variant has a step-loop kernel, 0 = always a graph of single launches
static std::atomic<int> g_fused_steps{1};

namespace caar {
struct LaunchChoice {
  int variant;
  int xcd_chunked;
  long long cache_window;
  bool adapted;  // the adaptive window's policy is already folded in (apply_window_policy): launch_with leaves it alone
};
static LaunchChoice launch_choice(const Config* cfg) {
  LaunchChoice c;
  c.variant = cfg ? cfg->selected.load(std::memory_order_relaxed) : 0;
  c.xcd_chunked = g_xcd_chunked.load(std::memory_order_relaxed);
  if (c.xcd_chunked < 0) c.xcd_chunked = cfg && cfg->variants[c.variant].prefers_xcd_chunked ? 1 : 0;
  c.cache_window = g_cache_window.load(std::memory_order_relaxed);
  c.adapted = false;
  return c;
}
}  // namespace caar

// ---- adaptive cache window --------------------------------------------------------------------------------------
// The hybrid policy pays when the accumulators it leaves in the Infinity Cache are still there at the next call: 88 % of
// the HBM peak (algorithmic) against 77 % all-streaming when the host replays the routine or alternates it with streaming
// kernels, but 75.9 against 76.7 % when a neighbour with the default cache policy moves a few hundred MB in between and
// evicts them (profiles/r04/adaptive_window_demo.log; bench.py roofline.interleaved).  What the host runs between two
// calls cannot be known here, so it is MEASURED: per array set, whole-range launches on a hybrid kernel get a HIP event
// recorded in front of them now and then (never waited for: polled with hipEventQuery at later calls), and the time from
// the start of one call to the start of the next is what a call costs the host under the policy it ran with — the kernel
// AND what it does to the neighbour that follows (under eviction the window-policy kernel itself is the faster one, 0.336
// against 0.351 ms, and the neighbour pays for writing its dirty lines back: timing the kernel alone picks the wrong side).
//   * After kFirstProbe calls, whenever the current policy's call-to-call time drifts up by more than kDrift (the host's
//     pattern changed), every kReprobe calls while the current policy is all-streaming (trying the window is cheap) and every
//     kReprobeWindow calls while it is the window,
//     the other policy runs for kWarm + kMeas calls, then the current one again for kWarm + kMeas, and the faster (medians
//     of the measured calls; the window on ties) becomes the policy.
//   * A launch inside a stream capture, on a sub-range of the elements, or of a captured graph uses the set's current
//     policy and measures nothing.
// Both policies give identical results (the all-streaming twin performs the same roundings).
//
// Locking (round 5): the policy of a set is PUBLISHED in atomics that launches read without a lock — sub-range launches
// (HOMME's horizontal-OpenMP threads), captures and graph re-validation only ever do that.  A whole-range launch also counts
// itself down on an atomic; WindowTuners::mu is taken only when the countdown says the state machine has work for this
// call: a passive sample is due (two consecutive calls in every kSampleEvery), a probe is running, or the set is new.
// Events are recorded and queried under the mutex only, so a reset / LRU restart can never destroy an event another thread
// is about to record.
namespace caar {
// One array set: the HIP-free state machine (caar_window_tuner.h) + what is published for lock-free readers + its HIP events.
struct WindowTuner {
  // published, read lock-free (a reader that races with a restart may see another set's policy for one call: both policies
  // compute the same thing)
  std::atomic<const void*> key{nullptr};  // CaarArrays::elem_derived_vn0 of the set
  std::atomic<int> device{-1};
  std::atomic<int> use_window{1};
  std::atomic<long long> idle{0};  // whole-range calls that may still pass without the mutex
  // everything below: under WindowTuners::mu
  WindowTunerState st;
  long long last_use = 0;
  hipEvent_t ev[WindowTunerState::kSlots] = {};

  void restart(const void* k, int dev) {  // (the events are kept for the next set: creating them is the expensive part)
    key.store(nullptr, std::memory_order_relaxed);
    st.reset();
    last_use = 0;
    use_window.store(1, std::memory_order_relaxed);
    idle.store(0, std::memory_order_relaxed);
    device.store(dev, std::memory_order_relaxed);
    key.store(k, std::memory_order_release);
  }
};
// the state machine's view of the tuner's HIP events on the launch stream
struct HipStampBackend {
  WindowTuner* t;
  hipStream_t stream;
  bool stamp(int slot) {
    hipEvent_t& e = t->ev[slot];
    if (!e && hipEventCreate(&e) != hipSuccess) e = nullptr;
    return e && hipEventRecord(e, stream) == hipSuccess;
  }
  bool ready(int slot) { return t->ev[slot] && hipEventQuery(t->ev[slot]) == hipSuccess; }
  bool elapsed(int a, int b, float* ms) {
    *ms = 0.f;
    return t->ev[a] && t->ev[b] && hipEventElapsedTime(ms, t->ev[a], t->ev[b]) == hipSuccess && *ms > 0.f;
  }
};
struct WindowTuners {
  std::mutex mu;
  WindowTuner t[8];
  long long tick = 0;
  std::atomic<long long> locked_calls{0};  // how often the mutex was taken on the launch path (tests, host_reentrancy.cpp)
};
static WindowTuners* window_tuners() {
  static WindowTuners* w = new WindowTuners();  // never destroyed: no HIP calls from static destructors at exit
  return w;
}
static WindowTuner* find_tuner(WindowTuners* all, const void* key, int device) {
  for (auto& x : all->t)
    if (x.key.load(std::memory_order_acquire) == key && x.device.load(std::memory_order_relaxed) == device) return &x;
  return nullptr;
}
// forget the set whose derived_vn0 lies at `key` (its memory is being released: a later allocation at the same address
// must not inherit its policy and counters); caar_arrays_free and caar_destroy call this
void window_tuner_forget(const void* key, int device) {
  if (!key) return;
  WindowTuners* all = window_tuners();
  std::lock_guard<std::mutex> g(all->mu);
  for (auto& x : all->t)
    if (x.key.load(std::memory_order_relaxed) == key && x.device.load(std::memory_order_relaxed) == device) x.restart(nullptr, -1);
}
}  // namespace caar
static std::atomic<int> g_adaptive_window{1};

// The policy in force for an array set (1 = window: also for a set no whole-range launch has been seen of).  Lock-free.
static int adaptive_window_current(const CaarArrays* dev, int device) {
  const caar::WindowTuner* t = caar::find_tuner(caar::window_tuners(), dev->elem_derived_vn0, device);
  return t ? t->use_window.load(std::memory_order_relaxed) : 1;
}

// Decides the policy of one whole-range launch on a hybrid kernel and, when the tuner wants a time stamp in front of this
// call, records it on `stream` (under the mutex).  Returns 1: use the window, 0: all streaming.
static int adaptive_window_policy(const CaarArrays* dev, int device, hipStream_t stream) {
  using caar::WindowTuner;
  caar::WindowTuners* all = caar::window_tuners();
  {
    // the common case: nothing is due for this call
    WindowTuner* t = caar::find_tuner(all, dev->elem_derived_vn0, device);
    if (t && t->idle.fetch_sub(1, std::memory_order_relaxed) > 0) return t->use_window.load(std::memory_order_relaxed);
  }
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
  all->locked_calls.fetch_add(1, std::memory_order_relaxed);
  std::lock_guard<std::mutex> g(all->mu);
  WindowTuner* t = nullptr;
  WindowTuner* lru = &all->t[0];
  for (auto& x : all->t) {
    if (x.key.load(std::memory_order_relaxed) == dev->elem_derived_vn0 && x.device.load(std::memory_order_relaxed) == device) t = &x;
    if (x.last_use < lru->last_use) lru = &x;
  }
  if (!t) {
    if (capturing) return 1;
    t = lru;
    t->restart(dev->elem_derived_vn0, device);
  }
  t->last_use = ++all->tick;
  if (capturing) return t->use_window.load(std::memory_order_relaxed);  // (idle stays <= 0: the next plain call comes here again)
  caar::HipStampBackend be{t, stream};
  const int policy = t->st.step(be);
  (void)hipGetLastError();  // hipEventQuery reports "not ready" as an error code
  t->use_window.store(t->st.use_window, std::memory_order_relaxed);
  t->idle.store(t->st.idle_granted, std::memory_order_relaxed);
  return policy;
}

// The cache window is a budget of the DEVICE: contexts that live on the same device share it in proportion to the bytes
// the hybrid policy could keep for each (vn0, omega_p, eta_dot_dpdn of every element).  One context gets all of it.
namespace caar {
struct DeviceShares {
  std::atomic<long long> keepable[64];  // per device: sum over live contexts of num_elems * keepable bytes per element
};
static DeviceShares* device_shares() {
  static DeviceShares* s = new DeviceShares();  // never destroyed (contexts may be released from finalisers at exit); zero-initialised
  return s;
}
static long long keepable_bytes(const CaarDims& d) {
  const long long pp = (long long)d.np * d.np;
  return (long long)d.num_elems * 8 * (4 * pp * d.nlev + pp);
}
static void device_shares_add(int device, long long bytes) {
  if (device < 0 || device >= 64) return;
  device_shares()->keepable[device].fetch_add(bytes, std::memory_order_relaxed);
}
// this context's part of `window` (lock-free: read on every caar_run)
static long long device_share_of(int device, const CaarDims& d, long long window) {
  if (device < 0 || device >= 64) return window;
  const long long mine = keepable_bytes(d), all = device_shares()->keepable[device].load(std::memory_order_relaxed);
  if (all <= mine || all <= 0) return window;
  return (long long)((double)window * ((double)mine / (double)all));
}
}  // namespace caar

struct CaarContext {
  CaarDims dims;
  int device;
  hipStream_t stream;
  CaarArrays dev;       // device pointers
  CaarArena* arena;     // what owns them (caar_arrays_alloc)
  caar::HostConstants consts;  // Dvv and hybi on the device
  double* norms_dev;    // 3 * num_elems
  double* stage_dev;    // staging for Fortran-ordered host arrays (largest array), lazily allocated
  // caar_run_steps: the captured graph and what it was captured for
  hipGraphExec_t steps_exec;
  CaarParams steps_params;
  caar::LaunchChoice steps_choice;  // variant / mapping / cache window baked into the captured launches
  int steps_n, steps_rotate;
  double steps_dvv[64];
  std::vector<double>* steps_hybi;
  bool shares_registered;  // counted in caar::device_shares()
};

static double** array_slot(CaarArrays* a, int i) { return reinterpret_cast<double**>(a) + i; }
static double* const* array_slot(const CaarArrays* a, int i) {
  return reinterpret_cast<double* const*>(a) + i;
}
static int array_ncomp(int i);
// the arrays compute_and_apply_rhs writes (SURVEY.md 8b "mutated in place")
static const int kMutated[] = {6, 7, 8, 11, 12, 13, 15};

#define HIP_TRY(expr)                     \
  do {                                    \
    hipError_t e_ = (expr);               \
    if (e_ != hipSuccess) return (int)e_; \
  } while (0)

extern "C" {

int caar_abi_version(void) { return CAAR_ABI_VERSION; }

int caar_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

long long caar_debug_dp3d_violations(int reset) {
#ifdef CAAR_DEBUG
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  const long long a = caar::debug_dp3d_count_np4(reset), b = caar::debug_dp3d_count_np4_steps(reset),
                  c = caar::debug_dp3d_count_np8(reset);
  return (a < 0 || b < 0 || c < 0) ? -2 : a + b + c;
#else
  (void)reset;
  return -1;  // not a debug build
#endif
}

int caar_supported(int np, int nlev) { return caar::find_config(np, nlev) != nullptr; }

int caar_supported_ex(int np, int nlev, int rsplit) {
  const caar::Config* c = caar::find_config(np, nlev);
  if (!c || rsplit < 0) return 0;
  if (rsplit > 0) return 1;
  // the Eulerian form: every specialised shape holds it; the run-time-level-count kernel (c->nlev == 0) up to 128 levels —
  // beyond that it spills 65-90 VGPRs and is compiled into libcaar_hip_extra.so only (caar_np4.hip launch_np4_dyn)
  return (c->nlev != 0 || nlev <= 128 || CAAR_EXTRA_NLEV) ? 1 : 0;
}

const char* caar_kernel_name(int np, int nlev) {
  const caar::Config* c = caar::find_config(np, nlev);
  return c ? c->variants[c->selected.load()].kernel : nullptr;
}

int caar_num_variants(int np, int nlev) {
  const caar::Config* c = caar::find_config(np, nlev);
  return c ? c->count : 0;
}

int caar_select_variant(int np, int nlev, int variant) {
  caar::Config* c = caar::find_config(np, nlev);
  if (!c) return CAAR_EUNSUPPORTED;
  if (variant < 0 || variant >= c->count) return CAAR_EINVAL;
  c->selected.store(variant);
  return CAAR_OK;
}

int caar_selected_variant(int np, int nlev) {
  const caar::Config* c = caar::find_config(np, nlev);
  return c ? c->selected.load() : CAAR_EUNSUPPORTED;
}

int caar_set_cache_window(long long bytes) {
  if (bytes < 0) return CAAR_EINVAL;
  g_cache_window.store(bytes);
  return CAAR_OK;
}

long long caar_get_cache_window(void) { return g_cache_window.load(); }

int caar_set_adaptive_window(int on) {
  g_adaptive_window.store(on ? 1 : 0);
  return CAAR_OK;
}
int caar_get_adaptive_window(void) { return g_adaptive_window.load(); }

int caar_adaptive_window_state(const double* vn0_dev, double* ms_window, double* ms_streaming, long long* probes) {
  caar::WindowTuners* all = caar::window_tuners();
  std::lock_guard<std::mutex> g(all->mu);
  for (auto& t : all->t)
    if (vn0_dev && t.key.load(std::memory_order_relaxed) == (const void*)vn0_dev) {
      if (ms_window) *ms_window = t.st.ms_window;
      if (ms_streaming) *ms_streaming = t.st.ms_streaming;
      if (probes) *probes = t.st.probes;
      return t.use_window.load(std::memory_order_relaxed);
    }
  return -1;
}

long long caar_adaptive_window_lock_count(void) { return caar::window_tuners()->locked_calls.load(std::memory_order_relaxed); }

int caar_adaptive_window_reset(void) {
  caar::WindowTuners* all = caar::window_tuners();
  std::lock_guard<std::mutex> g(all->mu);
  for (auto& t : all->t) t.restart(nullptr, -1);
  return CAAR_OK;
}

int caar_set_fused_steps(int on) {
  g_fused_steps.store(on ? 1 : 0);
  return CAAR_OK;
}

int caar_get_fused_steps(void) { return g_fused_steps.load(); }

int caar_has_fused_steps(int np, int nlev, int variant) {
  const caar::Config* c = caar::find_config(np, nlev);
  return (c && variant >= 0 && variant < c->count && c->variants[variant].launch_steps) ? 1 : 0;
}

int caar_set_xcd_chunked(int on) {
  g_xcd_chunked.store(on < 0 ? -1 : (on ? 1 : 0));
  return CAAR_OK;
}

const char* caar_variant_info(int np, int nlev, int variant) {
  const caar::Config* c = caar::find_config(np, nlev);
  return (c && variant >= 0 && variant < c->count) ? c->variants[variant].what : nullptr;
}

const char* caar_strerror(int rc) {
  switch (rc) {
    case CAAR_OK: return "ok";
    case CAAR_EINVAL: return "invalid argument";
    case CAAR_EUNSUPPORTED: return "no kernel compiled for this (np, nlev, rsplit) in this build of the library";
    case CAAR_ENODEVICE: return "no usable HIP device";
    case CAAR_ENOMEM: return "out of memory";
    default: return rc > 0 ? hipGetErrorString((hipError_t)rc) : "unknown caar error";
  }
}

long long caar_array_len(const CaarDims* d, int i) {
  if (!d || i < 0 || i >= CAAR_NUM_ARRAYS) return -1;
  const long long ne = d->num_elems, pp = (long long)d->np * d->np, blk = pp * d->nlev;
  switch (i) {
    case 0: case 1: return ne * pp * 4;
    case 2: case 3: case 4: case 5: case 9: return ne * pp;
    case 6: case 8: return ne * d->timelevels * blk;
    case 7: return ne * d->timelevels * blk * 2;
    case 10: return ne * d->qsize_d * 2 * blk;
    case 11: return ne * (blk + pp);
    case 12: case 13: case 14: return ne * blk;
    default: return ne * blk * 2;
  }
}

long long caar_algorithmic_bytes(int np, int nlev, int dry) {
  const long long pp = (long long)np * np;
  return 8 * ((dry ? 20 : 21) * pp * nlev + 2 * pp * (nlev + 1) + 13 * pp);
}

static int check_common(const CaarDims* d, const CaarParams* p) {
  if (!d || !p) return CAAR_EINVAL;
  if (d->num_elems < 0 || d->qsize_d < 1 || d->timelevels < 1) return CAAR_EINVAL;
  if (p->nets < 0 || p->nete > d->num_elems || p->nets > p->nete) return CAAR_EINVAL;
  const int tl = d->timelevels;
  if (p->n0 < 0 || p->n0 >= tl || p->np1 < 0 || p->np1 >= tl || p->nm1 < 0 || p->nm1 >= tl)
    return CAAR_EINVAL;
  if (p->qn0 < -1 || p->qn0 > 1) return CAAR_EINVAL;  // Qdp holds 2 time slots (data_structures.cpp:27)
  if (p->rsplit < 0) return CAAR_EINVAL;
  return CAAR_OK;
}

static void fill_args_impl(caar::KernelArgs& k, const CaarDims* dims, const CaarArrays* dev,
                           const double* dvv_dev, const CaarParams* p, const caar::LaunchChoice& ch);

static int launch_with(const CaarDims* dims, const CaarArrays* dev, const double* dvv_dev, const CaarParams* p,
                       void* stream, const caar::LaunchChoice* forced, bool tune = true);

// Is there a policy to decide for this launch?  Only where there is something to decide: a hybrid kernel with a window, the
// vertically Lagrangian form, and a data set that does not fit the 256 MB cache whole (one that does is served from it under
// either policy, and the small-element hosts that live in the launch-latency regime are not charged the tuner's microseconds).
static bool adaptive_applies(const caar::Config* cfg, const caar::LaunchChoice& ch, const CaarDims* dims, const CaarParams* p) {
  return cfg->variants[ch.variant].hybrid && ch.cache_window > 0 && p->rsplit != 0 && g_adaptive_window.load(std::memory_order_relaxed) &&
         caar_algorithmic_bytes(dims->np, dims->nlev, 0) * (long long)dims->num_elems > (256LL << 20);
}
// Bakes a policy into a launch choice: all-streaming = an empty window and, where the variant names one, the all-streaming
// kernel of the same shape with the element mapping it was measured faster with (unless the host fixed the mapping).
static void apply_window_policy(const caar::Config* cfg, caar::LaunchChoice& ch, int use_window) {
  ch.adapted = true;
  if (use_window) return;
  ch.cache_window = 0;
  const int twin = cfg->variants[ch.variant].streaming_twin;
  if (twin >= 0 && twin < cfg->count) {
    ch.variant = twin;
    if (g_xcd_chunked.load(std::memory_order_relaxed) < 0) ch.xcd_chunked = cfg->variants[twin].prefers_xcd_chunked ? 1 : 0;
  }
}

int caar_launch(const CaarDims* dims, const CaarArrays* dev, const double* dvv_dev,
                const CaarParams* p, void* stream) {
  return launch_with(dims, dev, dvv_dev, p, stream, nullptr);
}

// forced != nullptr: the tuning knobs the caller already read (a graph capture bakes ONE choice into all of
// its launches and into its cache key)
static int launch_with(const CaarDims* dims, const CaarArrays* dev, const double* dvv_dev, const CaarParams* p,
                       void* stream, const caar::LaunchChoice* forced, bool tune) {
  int rc = check_common(dims, p);
  if (rc) return rc;
  if (!dev || !dvv_dev) return CAAR_EINVAL;
  for (int i = 0; i < CAAR_NUM_ARRAYS; ++i)
    if (!*array_slot(dev, i)) return CAAR_EINVAL;
  // the kernels move v and vn0 as 16-byte (u, v) pairs and everything else as 8-byte doubles
  if (((size_t)dev->elem_state_v | (size_t)dev->elem_derived_vn0) & 15) return CAAR_EINVAL;
  for (int i = 0; i < CAAR_NUM_ARRAYS; ++i)
    if ((size_t)*array_slot(dev, i) & 7) return CAAR_EINVAL;
  const caar::Config* cfg = caar::find_config(dims->np, dims->nlev);
  if (!cfg) return CAAR_EUNSUPPORTED;
  if (!caar_supported_ex(dims->np, dims->nlev, p->rsplit)) return CAAR_EUNSUPPORTED;  // (before anything is enqueued)
  if (p->rsplit == 0 && (!p->hybi_dev || ((size_t)p->hybi_dev & 7))) return CAAR_EINVAL;
  const int n = p->nete - p->nets;
  if (n == 0) return CAAR_OK;

  caar::LaunchChoice ch = forced ? *forced : caar::launch_choice(cfg);  // the knobs, read once
  // the adaptive window: whole-range launches of a hybrid kernel with a window to switch off (see adaptive_window_policy)
  if (!ch.adapted && adaptive_applies(cfg, ch, dims, p)) {
    int device = 0;
    if (hipGetDevice(&device) == hipSuccess) {
      const bool measure = tune && n == dims->num_elems;  // else: the set's current policy (lock-free), nothing measured
      apply_window_policy(cfg, ch, measure ? adaptive_window_policy(dev, device, (hipStream_t)stream) : adaptive_window_current(dev, device));
    }
  }
  caar::KernelArgs k;
  fill_args_impl(k, dims, dev, dvv_dev, p, ch);
  const hipError_t e = cfg->variants[ch.variant].launch(k, n, (hipStream_t)stream);
  return e == hipErrorNotSupported ? CAAR_EUNSUPPORTED : (int)e;  // (belt and braces: caar_supported_ex above refuses first)
}

// nsteps calls as ONE launch if the chosen variant has a step-loop kernel (and the knob allows it): returns 1 if it was
// launched, 0 if the caller has to issue single launches, < 0 / a hipError_t on failure
static int try_fused_steps(const CaarDims* dims, const CaarArrays* dev, const double* dvv_dev, const CaarParams* p,
                           int nsteps, int rotate, void* stream, const caar::Config* cfg, const caar::LaunchChoice& ch,
                           int* rc_out) {
  *rc_out = CAAR_OK;
  // (a non-finite eta_ave_w: the step loops leave out the later calls' `eta_dot_dpdn += eta_ave_w * 0`, which is the
  // identity only where that product is a zero)
  // (nsteps == 1: one call is what the single launch is tuned for — hybrid cache window, XCD preference; the step-loop
  // kernels run the default or the all-streaming policy and lose 14 % on a lone call, profiles/r03/steps_bench_72_4w_final.log.
  // From two calls on the loop is ahead: profiles/r04/steps_bench_nsteps.log)
  if (!g_fused_steps.load(std::memory_order_relaxed) || nsteps < 2 || !cfg || !cfg->variants[ch.variant].launch_steps ||
      p->rsplit == 0 || !(p->eta_ave_w * 0.0 == 0.0))
    return 0;
  int rc = check_common(dims, p);
  if (rc == CAAR_OK && (!dev || !dvv_dev)) rc = CAAR_EINVAL;
  for (int i = 0; rc == CAAR_OK && i < CAAR_NUM_ARRAYS; ++i)
    if (!*array_slot(dev, i) || ((size_t)*array_slot(dev, i) & 7)) rc = CAAR_EINVAL;
  if (rc == CAAR_OK && (((size_t)dev->elem_state_v | (size_t)dev->elem_derived_vn0) & 15)) rc = CAAR_EINVAL;
  *rc_out = rc;
  if (rc != CAAR_OK) return 1;
  const int n = p->nete - p->nets;
  if (n == 0) return 1;
  caar::KernelArgs k;
  fill_args_impl(k, dims, dev, dvv_dev, p, ch);
  *rc_out = (int)cfg->variants[ch.variant].launch_steps(k, n, nsteps, rotate ? 1 : 0, (hipStream_t)stream);
  return 1;
}

int caar_launch_steps(const CaarDims* dims, const CaarArrays* dev, const double* dvv_dev, const CaarParams* p, int nsteps,
                      int rotate, void* stream) {
  if (!dims || !p || nsteps < 1) return CAAR_EINVAL;
  const caar::Config* cfg = caar::find_config(dims->np, dims->nlev);
  if (!cfg) return CAAR_EUNSUPPORTED;
  const caar::LaunchChoice ch = caar::launch_choice(cfg);
  int rc = CAAR_OK;
  if (try_fused_steps(dims, dev, dvv_dev, p, nsteps, rotate, stream, cfg, ch, &rc)) return rc;
  CaarParams q = *p;
  for (int i = 0; i < nsteps && rc == CAAR_OK; ++i) {
    rc = launch_with(dims, dev, dvv_dev, &q, stream, &ch);
    if (rotate) {  // data_structures.cpp:174-180
      const int t = q.np1;
      q.np1 = q.nm1;
      q.nm1 = q.n0;
      q.n0 = t;
    }
  }
  return rc;
}

static void fill_args_impl(caar::KernelArgs& k, const CaarDims* dims, const CaarArrays* dev,
                           const double* dvv_dev, const CaarParams* p, const caar::LaunchChoice& ch) {
  k.D = dev->elem_D;
  k.Dinv = dev->elem_Dinv;
  k.fcor = dev->elem_fcor;
  k.spheremp = dev->elem_spheremp;
  k.metdet = dev->elem_metdet;
  k.rmetdet = dev->elem_rmetdet;
  k.dp3d = dev->elem_state_dp3d;
  k.v = dev->elem_state_v;
  k.T = dev->elem_state_T;
  k.phis = dev->elem_state_phis;
  k.Qdp = dev->elem_state_Qdp;
  k.eta_dot_dpdn = dev->elem_derived_eta_dot_dpdn;
  k.omega_p = dev->elem_derived_omega_p;
  k.phi = dev->elem_derived_phi;
  k.pecnd = dev->elem_derived_pecnd;
  k.vn0 = dev->elem_derived_vn0;
  k.Dvv = dvv_dev;
  k.hybi = p->rsplit == 0 ? p->hybi_dev : nullptr;
  k.vadv = p->rsplit == 0 ? 1 : 0;
  k.nets = p->nets;
  k.nelem = p->nete - p->nets;
  k.per_xcd = ch.xcd_chunked ? (k.nelem + 7) / 8 : 0;
  {
    // what the hybrid policy keeps per chosen element: vn0 (2 blocks), omega_p, eta_dot_dpdn
    const long long pp = (long long)dims->np * dims->np;
    const long long per_elem = 8 * (4 * pp * dims->nlev + pp);
    const long long n = per_elem > 0 ? ch.cache_window / per_elem : 0;
    // that many of the arrays' elements, spread evenly over the WHOLE element range (a property of the element index,
    // element_is_cached): in steady state a constant share of the workgroups is served by the cache instead of HBM, and
    // a host that cuts the range into several launches — HOMME's horizontal OpenMP threads on disjoint [nets, nete)
    // (data_structures.hpp:58-69), one after the other or side by side on several streams — keeps the same set, i.e. ONE
    // window per device, not one per launch (round 3 budgeted per launch: four sub-range launches claimed 4 x 224 MB of
    // a 256 MB cache)
    const long long count = n <= 0 ? 0 : (n >= dims->num_elems ? dims->num_elems : n);
    // c / n in 32.32 fixed point (element_is_cached multiplies instead of dividing); c == n: exactly 2^32, every element
    const unsigned long long q = dims->num_elems > 0 ? (((unsigned long long)count << 32) / (unsigned long long)dims->num_elems) : 0ull;
    k.cache_q_lo = (unsigned)(q & 0xffffffffull);
    k.cache_q_hi = (unsigned)(q >> 32);
  }
  k.n0 = p->n0;
  k.np1 = p->np1;
  k.nm1 = p->nm1;
  k.qn0 = p->qn0;
  k.qsize_d = dims->qsize_d;
  k.timelevels = dims->timelevels;
  k.nlev = dims->nlev;
  k.dt2 = p->dt2;
  k.rrearth = p->rrearth;
  k.eta_ave_w = p->eta_ave_w;
  k.rv_over_rd_m1 = p->Rwater_vapor / p->Rgas - 1.0;  // P:151
  k.Rgas = p->Rgas;
  k.kappa = p->kappa;
  k.p_top = p->hyai0 * p->ps0;  // P:84
}

// gradient / divergence / vorticity_sphere of the path (codes 0..2) on the CaarArrays geometry: the same kernel
// as caar_sphere_operator_ex
static int three_operators(int np, int which, const CaarArrays* dev, const double* dvv_dev, int e0, int ne, int nlevels,
                           const double* in_dev, double* out_dev, double rrearth, void* stream) {
  caar::OpArgs a = {};
  a.D = dev->elem_D;
  a.Dinv = dev->elem_Dinv;
  a.metdet = dev->elem_metdet;
  a.rmetdet = dev->elem_rmetdet;
  a.dvv = dvv_dev;
  a.in = in_dev;
  a.out = out_dev;
  a.e0 = e0;
  a.ne = ne;
  a.nlevels = nlevels;
  a.rrearth = rrearth;
  return (int)caar::launch_sphere_operator_ex(np, which, a, (hipStream_t)stream);
}

int caar_sphere_operator(const CaarDims* dims, const CaarArrays* dev, const double* dvv_dev, int which, int ie,
                         int nlevels, const double* in_dev, double* out_dev, double rrearth, void* stream) {
  if (!dims || !dev || !dvv_dev || !in_dev || !out_dev || which < 0 || which > 2 || nlevels < 0) return CAAR_EINVAL;
  if (ie < 0 || ie >= dims->num_elems) return CAAR_EINVAL;
  if (dims->np != 4 && dims->np != 8) return CAAR_EUNSUPPORTED;
  if (!dev->elem_D || !dev->elem_Dinv || !dev->elem_metdet || !dev->elem_rmetdet) return CAAR_EINVAL;
  if ((((size_t)in_dev) | ((size_t)out_dev)) & 15) return CAAR_EINVAL;  // vector fields move as 16-byte (u, v) pairs
  return three_operators(dims->np, which, dev, dvv_dev, ie, 1, nlevels, in_dev, out_dev, rrearth, stream);
}

int caar_sphere_operator_range(const CaarDims* dims, const CaarArrays* dev, const double* dvv_dev, int which, int e0,
                               int e1, int nlevels, const double* in_dev, double* out_dev, double rrearth,
                               void* stream) {
  if (!dims || !dev || !dvv_dev || !in_dev || !out_dev || which < 0 || which > 2 || nlevels < 0) return CAAR_EINVAL;
  if (e0 < 0 || e1 > dims->num_elems || e0 > e1) return CAAR_EINVAL;
  if (dims->np != 4 && dims->np != 8) return CAAR_EUNSUPPORTED;
  if (!dev->elem_D || !dev->elem_Dinv || !dev->elem_metdet || !dev->elem_rmetdet) return CAAR_EINVAL;
  if ((((size_t)in_dev) | ((size_t)out_dev)) & 15) return CAAR_EINVAL;
  return three_operators(dims->np, which, dev, dvv_dev, e0, e1 - e0, nlevels, in_dev, out_dev, rrearth, stream);
}

int caar_sphere_operator_ex(const CaarDims* dims, const CaarOperatorGeometry* geo, const double* dvv_dev, int which,
                            int e0, int e1, int nlevels, const double* in_dev, double* out_dev,
                            const CaarOperatorScalars* sc, void* stream) {
  if (!dims || !geo || !dvv_dev || !out_dev || !sc || nlevels < 0) return CAAR_EINVAL;
  if (which < 0 || which >= CAAR_OP_COUNT) return CAAR_EINVAL;
  if (!in_dev && which != CAAR_OP_LAPLACE_TENSOR_REPLACE) return CAAR_EINVAL;  // the in-place form reads out_dev
  if (e0 < 0 || e1 > dims->num_elems || e0 > e1) return CAAR_EINVAL;
  if (dims->np != 4 && dims->np != 8) return CAAR_EUNSUPPORTED;
  if ((((size_t)in_dev) | ((size_t)out_dev)) & 15) return CAAR_EINVAL;
  const double* const* g = reinterpret_cast<const double* const*>(geo);
  const unsigned needs = caar::sphere_operator_ex_needs(which);
  for (int i = 0; i < 9; ++i) {
    if (!((needs >> i) & 1)) continue;
    if (!g[i]) return CAAR_EINVAL;
    const bool wide = i == 0 || i == 1 || i >= 6;  // D, Dinv, metinv, tensorVisc, vec_sph2cart move as 16-byte pairs
    if ((size_t)g[i] & (wide ? 15 : 7)) return CAAR_EINVAL;
  }
  caar::OpArgs a;
  a.D = geo->D; a.Dinv = geo->Dinv; a.metdet = geo->metdet; a.rmetdet = geo->rmetdet; a.spheremp = geo->spheremp;
  a.mp = geo->mp; a.metinv = geo->metinv; a.tensorVisc = geo->tensorVisc; a.vec_sph2cart = geo->vec_sph2cart;
  a.dvv = dvv_dev;
  a.in = in_dev;
  a.out = out_dev;
  a.e0 = e0;
  a.ne = e1 - e0;
  a.nlevels = nlevels;
  a.rrearth = sc->rrearth;
  a.alpha = sc->alpha;
  a.beta = sc->beta;
  a.nu_ratio = sc->nu_ratio;
  return (int)caar::launch_sphere_operator_ex(dims->np, which, a, (hipStream_t)stream);
}

int caar_euler_step(const CaarDims* dims, const CaarOperatorGeometry* geo, const double* dvv_dev, int e0, int e1,
                    int qsize, int qn0, double dt, double rrearth, const double* vstar_dev, const double* Qdp_dev,
                    double* qtens_dev, void* stream) {
  if (!dims || !geo || !dvv_dev || !vstar_dev || !Qdp_dev || !qtens_dev) return CAAR_EINVAL;
  if (e0 < 0 || e1 > dims->num_elems || e0 > e1) return CAAR_EINVAL;
  if (qsize < 0 || qsize > dims->qsize_d || qn0 < 0 || qn0 > 1 || dims->nlev < 1) return CAAR_EINVAL;
  if (dims->np != 4 && dims->np != 8) return CAAR_EUNSUPPORTED;
  if (!geo->Dinv || !geo->metdet || !geo->rmetdet) return CAAR_EINVAL;
  if ((((size_t)geo->Dinv) | ((size_t)vstar_dev)) & 15) return CAAR_EINVAL;  // move as 16-byte pairs
  if ((((size_t)geo->metdet) | ((size_t)geo->rmetdet) | ((size_t)Qdp_dev) | ((size_t)qtens_dev)) & 7) return CAAR_EINVAL;
  caar::EulerArgs a;
  a.Dinv = geo->Dinv; a.metdet = geo->metdet; a.rmetdet = geo->rmetdet; a.dvv = dvv_dev;
  a.vstar = vstar_dev; a.qdp = Qdp_dev; a.qtens = qtens_dev;
  a.e0 = e0; a.ne = e1 - e0; a.nlev = dims->nlev; a.qsize = qsize; a.qsize_d = dims->qsize_d; a.qn0 = qn0;
  a.dt = dt; a.rrearth = rrearth;
  return (int)caar::launch_euler_step(dims->np, a, (hipStream_t)stream);
}

int caar_preq_hydrostatic(const CaarDims* dims, int nelem, const double* phis_dev, const double* T_v_dev,
                          const double* p_dev, const double* dp_dev, double Rgas, double* phi_dev, void* stream) {
  if (!dims || nelem < 0 || !phis_dev || !T_v_dev || !p_dev || !dp_dev || !phi_dev) return CAAR_EINVAL;
  if (dims->np < 1 || dims->nlev < 2) return CAAR_EINVAL;
  return (int)caar::launch_preq_hydrostatic(dims->np, dims->nlev, nelem, phis_dev, T_v_dev, p_dev, dp_dev, Rgas, phi_dev,
                                            (hipStream_t)stream);
}

int caar_preq_omega_ps(const CaarDims* dims, int nelem, const double* p_dev, const double* vgrad_p_dev,
                       const double* divdp_dev, double* omega_p_dev, void* stream) {
  if (!dims || nelem < 0 || !p_dev || !vgrad_p_dev || !divdp_dev || !omega_p_dev) return CAAR_EINVAL;
  if (dims->np < 1 || dims->nlev < 2) return CAAR_EINVAL;
  return (int)caar::launch_preq_omega_ps(dims->np, dims->nlev, nelem, p_dev, vgrad_p_dev, divdp_dev, omega_p_dev,
                                         (hipStream_t)stream);
}

// ---- host-pointer convenience forms of the operators (one element, synchronous) -------------------------
namespace {
// One scratch area per device.  The host-pointer operators run on the calling thread's CURRENT device (a rank bound to
// GPU 3 stays on GPU 3) and never switch it.
struct HostOpScratch {
  std::mutex mu;
  hipStream_t stream = nullptr;
  double* dev = nullptr;
  size_t doubles = 0;
  // returns a device area of at least n doubles (grown on demand), creating the stream on first use
  hipError_t area(size_t n, double** out) {
    hipError_t e = hipSuccess;
    if (!stream) e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
    if (e == hipSuccess && doubles < n) {
      if (dev) (void)hipFree(dev);
      dev = nullptr;
      doubles = 0;
      e = hipMalloc((void**)&dev, sizeof(double) * n);
      if (e == hipSuccess) doubles = n;
    }
    *out = dev;
    return e;
  }
};
constexpr int kMaxDevices = 64;
HostOpScratch* host_op_scratch() {  // (a pointer: this sits inside the extern "C" block)
  static HostOpScratch* s = new HostOpScratch[kMaxDevices];  // never destroyed: no HIP calls from static destructors at exit
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  return s + dev;
}
}  // namespace

int caar_sphere_operator_host(const CaarDims* dims, const CaarArrays* host, const double* dvv_host, int which, int ie,
                              const double* in_host, double* out_host, double rrearth) {
  if (!dims || !host || !dvv_host || !in_host || !out_host || which < 0 || which > 2) return CAAR_EINVAL;
  if (ie < 0 || ie >= dims->num_elems) return CAAR_EINVAL;
  if (dims->np != 4 && dims->np != 8) return CAAR_EUNSUPPORTED;
  if (!host->elem_D || !host->elem_Dinv || !host->elem_metdet || !host->elem_rmetdet) return CAAR_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CAAR_ENODEVICE;
  const size_t pp = (size_t)dims->np * dims->np, nin = which == 0 ? pp : 2 * pp, nout = which == 0 ? 2 * pp : pp;
  HostOpScratch& s = *host_op_scratch();
  std::lock_guard<std::mutex> g(s.mu);
  double* d = nullptr;
  HIP_TRY(s.area(pp * 11 + nin + nout, &d));
  double *dD = d, *dDinv = dD + pp * 4, *dmet = dDinv + pp * 4, *drmet = dmet + pp, *ddvv = drmet + pp, *din = ddvv + pp,
         *dout = din + nin;
  HIP_TRY(hipMemcpyAsync(dD, host->elem_D + (size_t)ie * pp * 4, sizeof(double) * pp * 4, hipMemcpyHostToDevice, s.stream));
  HIP_TRY(hipMemcpyAsync(dDinv, host->elem_Dinv + (size_t)ie * pp * 4, sizeof(double) * pp * 4, hipMemcpyHostToDevice, s.stream));
  HIP_TRY(hipMemcpyAsync(dmet, host->elem_metdet + (size_t)ie * pp, sizeof(double) * pp, hipMemcpyHostToDevice, s.stream));
  HIP_TRY(hipMemcpyAsync(drmet, host->elem_rmetdet + (size_t)ie * pp, sizeof(double) * pp, hipMemcpyHostToDevice, s.stream));
  HIP_TRY(hipMemcpyAsync(ddvv, dvv_host, sizeof(double) * pp, hipMemcpyHostToDevice, s.stream));
  HIP_TRY(hipMemcpyAsync(din, in_host, sizeof(double) * nin, hipMemcpyHostToDevice, s.stream));
  caar::OpArgs a = {};
  a.D = dD;
  a.Dinv = dDinv;
  a.metdet = dmet;
  a.rmetdet = drmet;
  a.dvv = ddvv;
  a.in = din;
  a.out = dout;
  a.e0 = 0;
  a.ne = 1;
  a.nlevels = 1;
  a.rrearth = rrearth;
  HIP_TRY(caar::launch_sphere_operator_ex(dims->np, which, a, s.stream));
  HIP_TRY(hipMemcpyAsync(out_host, dout, sizeof(double) * nout, hipMemcpyDeviceToHost, s.stream));
  HIP_TRY(hipStreamSynchronize(s.stream));
  return CAAR_OK;
}

int caar_preq_hydrostatic_host(const CaarDims* dims, const double* phis, const double* T_v, const double* p, const double* dp,
                               double Rgas, double* phi) {
  if (!dims || !phis || !T_v || !p || !dp || !phi || dims->np < 1 || dims->nlev < 2) return CAAR_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CAAR_ENODEVICE;
  const size_t pp = (size_t)dims->np * dims->np, blk = pp * dims->nlev;
  HostOpScratch& s = *host_op_scratch();
  std::lock_guard<std::mutex> g(s.mu);
  double* d = nullptr;
  HIP_TRY(s.area(pp + 4 * blk, &d));
  double *d_phis = d, *d_Tv = d_phis + pp, *d_p = d_Tv + blk, *d_dp = d_p + blk, *d_phi = d_dp + blk;
  HIP_TRY(hipMemcpyAsync(d_phis, phis, sizeof(double) * pp, hipMemcpyHostToDevice, s.stream));
  HIP_TRY(hipMemcpyAsync(d_Tv, T_v, sizeof(double) * blk, hipMemcpyHostToDevice, s.stream));
  HIP_TRY(hipMemcpyAsync(d_p, p, sizeof(double) * blk, hipMemcpyHostToDevice, s.stream));
  HIP_TRY(hipMemcpyAsync(d_dp, dp, sizeof(double) * blk, hipMemcpyHostToDevice, s.stream));
  HIP_TRY(caar::launch_preq_hydrostatic(dims->np, dims->nlev, 1, d_phis, d_Tv, d_p, d_dp, Rgas, d_phi, s.stream));
  HIP_TRY(hipMemcpyAsync(phi, d_phi, sizeof(double) * blk, hipMemcpyDeviceToHost, s.stream));
  HIP_TRY(hipStreamSynchronize(s.stream));
  return CAAR_OK;
}

int caar_preq_omega_ps_host(const CaarDims* dims, const double* p, const double* vgrad_p, const double* divdp, double* omega_p) {
  if (!dims || !p || !vgrad_p || !divdp || !omega_p || dims->np < 1 || dims->nlev < 2) return CAAR_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CAAR_ENODEVICE;
  const size_t pp = (size_t)dims->np * dims->np, blk = pp * dims->nlev;
  HostOpScratch& s = *host_op_scratch();
  std::lock_guard<std::mutex> g(s.mu);
  double* d = nullptr;
  HIP_TRY(s.area(4 * blk, &d));
  double *d_p = d, *d_vg = d_p + blk, *d_dd = d_vg + blk, *d_om = d_dd + blk;
  HIP_TRY(hipMemcpyAsync(d_p, p, sizeof(double) * blk, hipMemcpyHostToDevice, s.stream));
  HIP_TRY(hipMemcpyAsync(d_vg, vgrad_p, sizeof(double) * blk, hipMemcpyHostToDevice, s.stream));
  HIP_TRY(hipMemcpyAsync(d_dd, divdp, sizeof(double) * blk, hipMemcpyHostToDevice, s.stream));
  HIP_TRY(caar::launch_preq_omega_ps(dims->np, dims->nlev, 1, d_p, d_vg, d_dd, d_om, s.stream));
  HIP_TRY(hipMemcpyAsync(omega_p, d_om, sizeof(double) * blk, hipMemcpyDeviceToHost, s.stream));
  HIP_TRY(hipStreamSynchronize(s.stream));
  return CAAR_OK;
}

int caar_reciprocal(const double* in_dev, double* out_dev, long long n, void* stream) {
  if (!in_dev || !out_dev || n < 0) return CAAR_EINVAL;
  return (int)caar::launch_reciprocal(in_dev, out_dev, (size_t)n, (hipStream_t)stream);
}

// components per GLL point of array i (CaarArrays member order)
static int array_ncomp(int i) { return (i == 0 || i == 1) ? 4 : ((i == 7 || i == 15) ? 2 : 1); }

static int convert_layout(const CaarDims* d, const CaarArrays* from, const CaarArrays* to, int e0, int e1,
                          bool to_caar, bool mutated_only, void* stream) {
  if (!d || !from || !to || e0 < 0 || e1 > d->num_elems || e0 > e1) return CAAR_EINVAL;
  if (d->np != 4 && d->np != 8) return CAAR_EUNSUPPORTED;
  double* dst[CAAR_NUM_ARRAYS];
  const double* src[CAAR_NUM_ARRAYS];
  size_t n[CAAR_NUM_ARRAYS];
  int nc[CAAR_NUM_ARRAYS], qo[CAAR_NUM_ARRAYS], count = 0;
  for (int i = 0; i < CAAR_NUM_ARRAYS; ++i) {
    bool wanted = !mutated_only;
    for (int m : kMutated) wanted = wanted || m == i;
    if (!wanted) continue;
    const double* s = *array_slot(from, i);
    double* t = *array_slot(to, i);
    if (!s || !t) return CAAR_EINVAL;
    const long long per = caar_array_len(d, i) / (d->num_elems ? d->num_elems : 1);
    // every array is element-major on both sides, so [e0, e1) is one contiguous range
    dst[count] = t + (size_t)per * e0;
    src[count] = s + (size_t)per * e0;
    n[count] = (size_t)per * (e1 - e0);
    nc[count] = array_ncomp(i);
    qo[count] = i == 10;
    ++count;
  }
  // all arrays of the conversion in one launch (caar_layout.hip)
  return (int)caar::launch_layout_all(d->np, count, dst, src, n, nc, qo, d->nlev, d->qsize_d, to_caar, (hipStream_t)stream);
}

int caar_layout_from_f90(const CaarDims* dims, const CaarArrays* f90_dev, const CaarArrays* caar_dev, int e0,
                         int e1, void* stream) {
  return convert_layout(dims, f90_dev, caar_dev, e0, e1, true, false, stream);
}

int caar_layout_to_f90(const CaarDims* dims, const CaarArrays* caar_dev, const CaarArrays* f90_dev, int e0,
                       int e1, int all_arrays, void* stream) {
  return convert_layout(dims, caar_dev, f90_dev, e0, e1, false, !all_arrays, stream);
}

/* Measurement utilities (roofline context; not used by the product path). */
int caar_stream_copy(double* dst_dev, const double* src_dev, long long n_doubles, int lane_bytes,
                     void* stream) {
  if (!dst_dev || !src_dev || n_doubles <= 0 || (lane_bytes != 8 && lane_bytes != 16)) return CAAR_EINVAL;
  return (int)caar::launch_stream_copy(dst_dev, src_dev, (size_t)n_doubles, lane_bytes, (hipStream_t)stream);
}

int caar_stream_copy_tuned(double* dst_dev, const double* src_dev, long long n_doubles, int variant, void* stream) {
  if (!dst_dev || !src_dev || n_doubles <= 0 || (n_doubles & 1)) return CAAR_EINVAL;
  if ((((size_t)dst_dev) | ((size_t)src_dev)) & 15) return CAAR_EINVAL;
  if (variant < 0 || variant >= caar::stream_copy_tuned_variants()) return CAAR_EINVAL;
  return (int)caar::launch_stream_copy_tuned(dst_dev, src_dev, (size_t)n_doubles, variant, (hipStream_t)stream);
}

int caar_stream_copy_tuned_variants(void) { return caar::stream_copy_tuned_variants(); }

const char* caar_stream_copy_tuned_info(int variant) { return caar::stream_copy_tuned_info(variant); }

int caar_traffic_skeleton(const CaarDims* dims, const CaarArrays* dev, const CaarParams* p, int variant,
                          void* stream) {
  int rc = check_common(dims, p);
  if (rc) return rc;
  if (!dev || (dims->np != 4 && dims->np != 8)) return CAAR_EUNSUPPORTED;
  for (int i = 0; i < CAAR_NUM_ARRAYS; ++i)
    if (!*array_slot(dev, i)) return CAAR_EINVAL;
  if (p->nete == p->nets) return CAAR_OK;
  caar::KernelArgs k;
  fill_args_impl(k, dims, dev, nullptr, p, caar::launch_choice(nullptr));
  if (dims->np == 8)
    return (int)caar::launch_traffic_skeleton_np8(k, dims->nlev, variant, p->nete - p->nets, (hipStream_t)stream);
  return (int)caar::launch_traffic_skeleton(k, dims->nlev, variant, p->nete - p->nets, (hipStream_t)stream);
}

int caar_launch_state_norms(const CaarDims* d, const CaarArrays* dev, int tl, int e0, int e1,
                            double* out_dev, void* stream) {
  if (!d || !dev || !out_dev || tl < 0 || tl >= d->timelevels || e0 < 0 || e1 > d->num_elems || e0 > e1)
    return CAAR_EINVAL;
  if (!dev->elem_state_v || !dev->elem_state_T || !dev->elem_state_dp3d) return CAAR_EINVAL;
  return (int)caar::launch_state_norms(dev->elem_state_v, dev->elem_state_T, dev->elem_state_dp3d, d->np,
                                       d->nlev, d->timelevels, tl, e0, e1, out_dev, (hipStream_t)stream);
}

// ------------------------------------------------------------------ context API
int caar_create(CaarContext** out, const CaarDims* dims, int device) { return caar_create_ex(out, dims, device, nullptr); }

int caar_create_ex(CaarContext** out, const CaarDims* dims, int device, const CaarPlacement* placement) {
  if (!out || !dims || dims->num_elems <= 0) return CAAR_EINVAL;
  if (!caar::find_config(dims->np, dims->nlev)) return CAAR_EUNSUPPORTED;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CAAR_ENODEVICE;
  if (device < 0 || device >= ndev) return CAAR_EINVAL;
  HIP_TRY(hipSetDevice(device));
  CaarContext* c = new (std::nothrow) CaarContext();
  if (!c) return CAAR_ENOMEM;
  std::memset(c, 0, sizeof(*c));
  c->dims = *dims;
  c->device = device;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) {
    const int rc = caar_arrays_alloc_ex(&c->arena, dims, device, placement, &c->dev);  // placed for bandwidth (caar_alloc.hip)
    if (rc != CAAR_OK) {
      caar_destroy(c);
      return rc;
    }
  }
  if (e == hipSuccess) e = c->consts.create(dims->nlev);
  if (e == hipSuccess) e = hipMalloc((void**)&c->norms_dev, sizeof(double) * 3 * dims->num_elems);
  if (e != hipSuccess) {
    caar_destroy(c);
    return e == hipErrorOutOfMemory ? CAAR_ENOMEM : (int)e;
  }
  caar::device_shares_add(device, caar::keepable_bytes(c->dims));
  c->shares_registered = true;
  *out = c;
  return CAAR_OK;
}

long long caar_context_cache_window(CaarContext* c) {
  if (!c) return -1;
  return caar::device_share_of(c->device, c->dims, g_cache_window.load(std::memory_order_relaxed));
}

void caar_destroy(CaarContext* c) {
  if (!c) return;
  if (c->shares_registered) caar::device_shares_add(c->device, -caar::keepable_bytes(c->dims));
  int caller_dev = -1;  // may run from a finaliser: the calling thread's current device is left as it was
  if (hipGetDevice(&caller_dev) != hipSuccess) caller_dev = -1;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->arena) (void)caar_arrays_free(c->arena);
  c->consts.destroy();
  if (c->norms_dev) (void)hipFree(c->norms_dev);
  if (c->stage_dev) (void)hipFree(c->stage_dev);
  if (c->steps_exec) (void)hipGraphExecDestroy(c->steps_exec);
  delete c->steps_hybi;
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (caller_dev >= 0 && caller_dev != c->device) (void)hipSetDevice(caller_dev);
  delete c;
}

static int copy_range(CaarContext* c, const CaarArrays* host, int i, int e0, int e1, bool to_device) {
  const long long per = caar_array_len(&c->dims, i) / c->dims.num_elems;
  double* h = *array_slot(host, i);
  double* d = *array_slot(&c->dev, i);
  if (!h) return CAAR_EINVAL;
  const size_t off = (size_t)per * e0, bytes = sizeof(double) * (size_t)per * (e1 - e0);
  if (to_device) HIP_TRY(hipMemcpyAsync(d + off, h + off, bytes, hipMemcpyHostToDevice, c->stream));
  else HIP_TRY(hipMemcpyAsync(h + off, d + off, bytes, hipMemcpyDeviceToHost, c->stream));
  return CAAR_OK;
}

int caar_upload(CaarContext* c, const CaarArrays* host, int e0, int e1) {
  if (!c || !host || e0 < 0 || e1 > c->dims.num_elems || e0 > e1) return CAAR_EINVAL;
  HIP_TRY(hipSetDevice(c->device));
  for (int i = 0; i < CAAR_NUM_ARRAYS; ++i) {
    int rc = copy_range(c, host, i, e0, e1, true);
    if (rc) return rc;
  }
  return CAAR_OK;
}

int caar_download(CaarContext* c, const CaarArrays* host, int e0, int e1, int all_arrays) {
  if (!c || !host || e0 < 0 || e1 > c->dims.num_elems || e0 > e1) return CAAR_EINVAL;
  HIP_TRY(hipSetDevice(c->device));
  if (all_arrays) {
    for (int i = 0; i < CAAR_NUM_ARRAYS; ++i) {
      int rc = copy_range(c, host, i, e0, e1, false);
      if (rc) return rc;
    }
  } else {
    for (int i : kMutated) {
      int rc = copy_range(c, host, i, e0, e1, false);
      if (rc) return rc;
    }
  }
  return CAAR_OK;
}

// Fortran-ordered HOST arrays <-> the context's device arrays: each array goes through one
// device staging buffer (H2D copy + layout kernel, or layout kernel + D2H copy), all ordered
// on the context stream, so the staging buffer is reused array after array.
static int f90_transfer(CaarContext* c, const CaarArrays* f90_host, int e0, int e1, bool upload, bool mutated_only,
                        unsigned mask = 0xffffu) {
  if (!c || !f90_host || e0 < 0 || e1 > c->dims.num_elems || e0 > e1) return CAAR_EINVAL;
  HIP_TRY(hipSetDevice(c->device));
  if (!c->stage_dev) {
    long long biggest = 0;
    for (int i = 0; i < CAAR_NUM_ARRAYS; ++i)
      if (caar_array_len(&c->dims, i) > biggest) biggest = caar_array_len(&c->dims, i);
    hipError_t e = hipMalloc((void**)&c->stage_dev, sizeof(double) * (size_t)biggest);
    if (e != hipSuccess) return e == hipErrorOutOfMemory ? CAAR_ENOMEM : (int)e;
  }
  for (int i = 0; i < CAAR_NUM_ARRAYS; ++i) {
    bool wanted = !mutated_only;
    for (int m : kMutated) wanted = wanted || m == i;
    if (!wanted || !((mask >> i) & 1u)) continue;
    double* h = *array_slot(f90_host, i);
    if (!h) return CAAR_EINVAL;
    const long long per = caar_array_len(&c->dims, i) / c->dims.num_elems;
    const size_t off = (size_t)per * e0, n = (size_t)per * (e1 - e0);
    if (n == 0) continue;
    double* d = *array_slot(&c->dev, i) + off;
    if (upload) {
      HIP_TRY(hipMemcpyAsync(c->stage_dev, h + off, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(caar::launch_layout(d, c->stage_dev, n, c->dims.np, array_ncomp(i), c->dims.nlev, c->dims.qsize_d,
                                  i == 10, true, c->stream));
    } else {
      HIP_TRY(caar::launch_layout(c->stage_dev, d, n, c->dims.np, array_ncomp(i), c->dims.nlev, c->dims.qsize_d,
                                  i == 10, false, c->stream));
      HIP_TRY(hipMemcpyAsync(h + off, c->stage_dev, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    }
  }
  return CAAR_OK;
}

int caar_upload_f90(CaarContext* c, const CaarArrays* f90_host, int e0, int e1) {
  return f90_transfer(c, f90_host, e0, e1, true, false);
}

int caar_upload_f90_arrays(CaarContext* c, const CaarArrays* f90_host, int e0, int e1, unsigned array_mask) {
  return f90_transfer(c, f90_host, e0, e1, true, false, array_mask & 0xffffu);
}

int caar_download_f90(CaarContext* c, const CaarArrays* f90_host, int e0, int e1, int all_arrays) {
  return f90_transfer(c, f90_host, e0, e1, false, !all_arrays);
}

int caar_run(CaarContext* c, const CaarParams* p) {
  if (!c || !p || !p->Dvv) return CAAR_EINVAL;
  HIP_TRY(hipSetDevice(c->device));
  if (p->rsplit == 0 && !p->hybi) return CAAR_EINVAL;
  if (!caar_supported_ex(c->dims.np, c->dims.nlev, p->rsplit)) return CAAR_EUNSUPPORTED;
  CaarParams q = *p;
  const double* dvv_dev = nullptr;
  HIP_TRY(c->consts.sync(c->dims, &q, c->stream, &dvv_dev));
  caar::LaunchChoice ch = caar::launch_choice(caar::find_config(c->dims.np, c->dims.nlev));
  ch.cache_window = caar::device_share_of(c->device, c->dims, ch.cache_window);  // the device's window, shared by its contexts
  return launch_with(&c->dims, &c->dev, dvv_dev, &q, c->stream, &ch);
}

int caar_run_steps(CaarContext* c, const CaarParams* p, int nsteps, int rotate) {
  if (!c || !p || !p->Dvv || nsteps < 1) return CAAR_EINVAL;
  if (p->rsplit == 0 && !p->hybi) return CAAR_EINVAL;
  if (!caar_supported_ex(c->dims.np, c->dims.nlev, p->rsplit)) return CAAR_EUNSUPPORTED;  // no capture is started
  HIP_TRY(hipSetDevice(c->device));
  CaarParams q = *p;
  const double* dvv_dev = nullptr;
  HIP_TRY(c->consts.sync(c->dims, &q, c->stream, &dvv_dev));  // uploads stay outside the graph
  const size_t nd = sizeof(double) * c->dims.np * c->dims.np, nh = (size_t)c->dims.nlev + 1;
  // the cached graph is valid for the same scalars AND the same Dvv / hybi VALUES (they are
  // baked into device buffers the kernels read, so only their content matters)
  // ... and for the same variant / element mapping / cache window, which fill_args bakes into the
  // captured kernel arguments (a knob changed after the capture must not replay the old launches)
  const caar::Config* cfg = caar::find_config(c->dims.np, c->dims.nlev);
  caar::LaunchChoice now = caar::launch_choice(cfg);
  now.cache_window = caar::device_share_of(c->device, c->dims, now.cache_window);
  {
    // one launch: every workgroup makes all nsteps calls for its element (caar_np4_steps_kernel)
    int rc = CAAR_OK;
    if (try_fused_steps(&c->dims, &c->dev, dvv_dev, &q, nsteps, rotate, c->stream, cfg, now, &rc)) return rc;
  }
  // ... and for the same cache POLICY: the adaptive window's current choice for this array set is folded into `now` (lock-free
  // read), so a policy the tuner has flipped since the capture re-captures the graph instead of replaying the old launches
  // for ever.  (Graph launches themselves measure nothing: a set driven only through this path keeps the default policy.)
  if (cfg && adaptive_applies(cfg, now, &c->dims, &q)) apply_window_policy(cfg, now, adaptive_window_current(&c->dev, c->device));
  bool same = c->steps_exec && c->steps_n == nsteps && c->steps_rotate == (rotate != 0) &&
              c->steps_choice.variant == now.variant && c->steps_choice.xcd_chunked == now.xcd_chunked &&
              c->steps_choice.cache_window == now.cache_window && std::memcmp(c->steps_dvv, p->Dvv, nd) == 0;
  if (same) {
    // field by field: the struct has padding bytes a caller need not have initialised
    const CaarParams &a = c->steps_params, &b = *p;
    same = a.nets == b.nets && a.nete == b.nete && a.n0 == b.n0 && a.np1 == b.np1 && a.nm1 == b.nm1 &&
           a.qn0 == b.qn0 && a.rsplit == b.rsplit &&
           std::memcmp(&a.dt2, &b.dt2, sizeof(double)) == 0 && std::memcmp(&a.rrearth, &b.rrearth, sizeof(double)) == 0 &&
           std::memcmp(&a.eta_ave_w, &b.eta_ave_w, sizeof(double)) == 0 &&
           std::memcmp(&a.Rwater_vapor, &b.Rwater_vapor, sizeof(double)) == 0 &&
           std::memcmp(&a.Rgas, &b.Rgas, sizeof(double)) == 0 && std::memcmp(&a.kappa, &b.kappa, sizeof(double)) == 0 &&
           std::memcmp(&a.ps0, &b.ps0, sizeof(double)) == 0 && std::memcmp(&a.hyai0, &b.hyai0, sizeof(double)) == 0;
  }
  if (same && p->rsplit == 0)
    same = c->steps_hybi && std::memcmp(c->steps_hybi->data(), p->hybi, sizeof(double) * nh) == 0;
  if (!same) {
    int rc = check_common(&c->dims, p);
    if (rc) return rc;
    if (c->steps_exec) {
      (void)hipGraphExecDestroy(c->steps_exec);
      c->steps_exec = nullptr;
    }
    hipGraph_t graph = nullptr;
    HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nsteps && rc == CAAR_OK; ++i) {
      rc = launch_with(&c->dims, &c->dev, dvv_dev, &q, c->stream, &now);
      if (rotate) {  // data_structures.cpp:174-180
        const int t = q.np1;
        q.np1 = q.nm1;
        q.nm1 = q.n0;
        q.n0 = t;
      }
    }
    hipError_t e = hipStreamEndCapture(c->stream, &graph);  // always end the capture, also after an error
    if (rc != CAAR_OK) {
      if (graph) (void)hipGraphDestroy(graph);
      return rc;
    }
    if (e != hipSuccess) return (int)e;
    e = hipGraphInstantiate(&c->steps_exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) {
      c->steps_exec = nullptr;
      return (int)e;
    }
    c->steps_params = *p;
    c->steps_choice = now;
    c->steps_n = nsteps;
    c->steps_rotate = rotate != 0;
    std::memcpy(c->steps_dvv, p->Dvv, nd);
    if (p->rsplit == 0) {
      if (!c->steps_hybi) c->steps_hybi = new (std::nothrow) std::vector<double>();
      if (!c->steps_hybi) return CAAR_ENOMEM;
      c->steps_hybi->assign(p->hybi, p->hybi + nh);
    }
  }
  HIP_TRY(hipGraphLaunch(c->steps_exec, c->stream));
  return CAAR_OK;
}

int caar_sync(CaarContext* c) {
  if (!c) return CAAR_EINVAL;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return CAAR_OK;
}

int caar_device_arrays(CaarContext* c, CaarArrays* out) {
  if (!c || !out) return CAAR_EINVAL;
  *out = c->dev;
  return CAAR_OK;
}

void* caar_stream(CaarContext* c) { return c ? (void*)c->stream : nullptr; }

int caar_state_norms(CaarContext* c, int tl, int e0, int e1, double out[3]) {
  if (!c || !out || tl < 0 || tl >= c->dims.timelevels || e0 < 0 || e1 > c->dims.num_elems || e0 > e1)
    return CAAR_EINVAL;
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(caar::launch_state_norms(c->dev.elem_state_v, c->dev.elem_state_T, c->dev.elem_state_dp3d,
                                   c->dims.np, c->dims.nlev, c->dims.timelevels, tl, e0, e1,
                                   c->norms_dev, c->stream));
  std::vector<double> h((size_t)3 * (e1 - e0));
  HIP_TRY(hipMemcpyAsync(h.data(), c->norms_dev, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  // P:388-390: vnorm += pow(compute_norm(...), 2) over elements, then sqrt (P:394-396);
  // the kernel stores pow(compute_norm, 2) per element and field.
  double s[3] = {0, 0, 0};
  for (int e = 0; e < e1 - e0; ++e)
    for (int f = 0; f < 3; ++f) s[f] += h[(size_t)3 * e + f];
  for (int f = 0; f < 3; ++f) out[f] = __builtin_sqrt(s[f]);
  return CAAR_OK;
}

int caar_time_runs(CaarContext* c, const CaarParams* p, int reps, float* ms_total) {
  if (!c || !p || reps <= 0 || !ms_total) return CAAR_EINVAL;
  HIP_TRY(hipSetDevice(c->device));
  hipEvent_t a, b;
  HIP_TRY(hipEventCreate(&a));
  HIP_TRY(hipEventCreate(&b));
  int rc = caar_run(c, p);  // warm-up (also uploads Dvv)
  if (rc == CAAR_OK) rc = (int)hipEventRecord(a, c->stream);
  for (int i = 0; rc == CAAR_OK && i < reps; ++i) rc = caar_run(c, p);
  if (rc == CAAR_OK) rc = (int)hipEventRecord(b, c->stream);
  if (rc == CAAR_OK) rc = (int)hipEventSynchronize(b);
  if (rc == CAAR_OK) rc = (int)hipEventElapsedTime(ms_total, a, b);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return rc;
}

// ------------------------------------------------------------------ host-mapped API
struct CaarHostMapping {
  CaarDims dims;
  int device;
  hipStream_t stream;
  CaarArrays host;       // what was registered (for hipHostUnregister)
  bool registered[CAAR_NUM_ARRAYS];
  CaarArrays dev;        // the same memory as the device addresses it
  caar::HostConstants consts;
  std::mutex* lock;      // caar_run_mapped from several host threads (disjoint [nets, nete)): SURVEY 8b
};

int caar_unmap_host(CaarHostMapping* m) {
  if (!m) return CAAR_EINVAL;
  (void)hipSetDevice(m->device);
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  int rc = CAAR_OK;
  for (int i = 0; i < CAAR_NUM_ARRAYS; ++i)
    if (m->registered[i]) {
      hipError_t e = hipHostUnregister(*array_slot(&m->host, i));
      if (e != hipSuccess && rc == CAAR_OK) rc = (int)e;
    }
  m->consts.destroy();
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m->lock;
  delete m;
  return rc;
}

int caar_map_host(CaarHostMapping** out, const CaarDims* dims, const CaarArrays* host, int device) {
  if (!out || !dims || !host || dims->num_elems <= 0 || dims->qsize_d < 1 || dims->timelevels < 1)
    return CAAR_EINVAL;
  if (!caar::find_config(dims->np, dims->nlev)) return CAAR_EUNSUPPORTED;
  for (int i = 0; i < CAAR_NUM_ARRAYS; ++i)
    if (!*array_slot(host, i)) return CAAR_EINVAL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return CAAR_ENODEVICE;
  if (device < 0 || device >= ndev) return CAAR_EINVAL;
  HIP_TRY(hipSetDevice(device));
  CaarHostMapping* m = new (std::nothrow) CaarHostMapping();
  if (!m) return CAAR_ENOMEM;
  std::memset(m, 0, sizeof(*m));
  m->dims = *dims;
  m->device = device;
  m->host = *host;
  m->lock = new (std::nothrow) std::mutex();
  if (!m->lock) {
    delete m;
    return CAAR_ENOMEM;
  }
  hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
  for (int i = 0; e == hipSuccess && i < CAAR_NUM_ARRAYS; ++i) {
    double* h = *array_slot(host, i);
    e = hipHostRegister(h, sizeof(double) * (size_t)caar_array_len(dims, i), hipHostRegisterMapped);
    if (e == hipSuccess) m->registered[i] = true;
    if (e == hipErrorHostMemoryAlreadyRegistered) {  // the host pinned it itself: use it as it is
      (void)hipGetLastError();
      e = hipSuccess;
    }
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)array_slot(&m->dev, i), h, 0);
  }
  if (e == hipSuccess) e = m->consts.create(dims->nlev);
  if (e != hipSuccess) {
    (void)caar_unmap_host(m);
    return e == hipErrorOutOfMemory ? CAAR_ENOMEM : (int)e;
  }
  *out = m;
  return CAAR_OK;
}

int caar_run_mapped(CaarHostMapping* m, const CaarParams* p) {
  if (!m || !p || !p->Dvv) return CAAR_EINVAL;
  HIP_TRY(hipSetDevice(m->device));
  if (p->rsplit == 0 && !p->hybi) return CAAR_EINVAL;
  if (!caar_supported_ex(m->dims.np, m->dims.nlev, p->rsplit)) return CAAR_EUNSUPPORTED;
  CaarParams q = *p;
  hipEvent_t done = nullptr;
  HIP_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  int rc;
  {
    // Re-entrant like the reference (disjoint [nets, nete) from several host threads with their own
    // Control copies): the constants cache and the enqueue are serialised, the wait is not — each caller
    // waits for its own launch only.
    std::lock_guard<std::mutex> g(*m->lock);
    const double* dvv_dev = nullptr;
    rc = (int)m->consts.sync(m->dims, &q, m->stream, &dvv_dev);
    if (rc == CAAR_OK) rc = caar_launch(&m->dims, &m->dev, dvv_dev, &q, m->stream);
    if (rc == CAAR_OK) rc = (int)hipEventRecord(done, m->stream);
  }
  if (rc == CAAR_OK) rc = (int)hipEventSynchronize(done);
  (void)hipEventDestroy(done);
  return rc;
}

}  // extern "C"
