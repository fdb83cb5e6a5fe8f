// homme_caar.hpp — the reference's call surface for the hot path, on the MI355X.
//
//   Homme::compute_and_apply_rhs(TestData&)      drop-in for the reference's
//       cxx/pointers_only/compute_and_apply_rhs.hpp:9 (host arrays in, host arrays out:
//       upload -> HIP kernel -> download of the mutated arrays), and the three helpers
//       its main.cpp also links (compute_norm :19, print_results_2norm :21,
//       dump_results_to_file :23);
//   Homme::DeviceSession                          the same call with the element arrays
//       kept resident on the GPU between calls (what a time-stepping host wants, and
//       what the driver's timed loop uses).
//
// Build with -DCAAR_USE_REFERENCE_HEADERS and the reference's include path to compile
// this file against the reference's own data_structures.hpp (link-level drop-in for the
// reference's main.cpp); otherwise homme_data.hpp is used.
#ifndef HOMME_CAAR_HPP
#define HOMME_CAAR_HPP

#ifdef CAAR_USE_REFERENCE_HEADERS
#include "data_structures.hpp"
#else
#include "homme_data.hpp"
#endif

struct CaarContext;

namespace Homme {

// Two modes (homme_caar.cpp): mapped, the default — the kernel works on the host's page-locked arrays over PCIe and the
// host may read them after every call; resident (environment CAAR_SHIM_RESIDENT=1, or -DCAAR_SHIM_RESIDENT) — the arrays
// are uploaded on the first call and later calls only enqueue the kernel: the host's copies of the mutated arrays are
// stale until sync_to_host(); print_results_2norm / dump_results_to_file / release_host_mapping know.
void compute_and_apply_rhs(TestData& data);
// Releases what compute_and_apply_rhs(TestData&) holds on the host arrays it ran on last (page locks; in resident mode the
// device copy, after writing its results back).  Call it before freeing those arrays (not part of the reference's
// surface: its callee holds no state).
void release_host_mapping();
// Resident mode: bring the host arrays up to date (the seven arrays the path mutates) / send the host's arrays to the
// device again after the host changed them (in that order: sync_to_host, change, sync_to_device — sync_to_device on a stale
// host copy aborts rather than undo the calls made since).  No-ops in mapped mode and for arrays the shim holds no device copy of.
// (cf. sync_to_host / sync_to_device of the reference's Kokkos variants, level_vectorized_ppscan/Utility.hpp)
void sync_to_host(TestData& data);
void sync_to_device(const TestData& data);
// What the calls through compute_and_apply_rhs(TestData&) cost so far: `seconds` is wall time on the host from the first
// enqueue after a wait to the completion of the last call (resident), or the sum of the calls' durations (mapped);
// page-locking, the initial upload and downloads are not in it.  Waits for the device.
struct ShimStats {
  int resident;
  long long calls;
  double seconds;
};
ShimStats shim_stats();
// compute_and_apply_rhs.hpp:11-17
void preq_hydrostatic(const real* const phis, const real* const T_v, const real* const p, const real* dp, real Rgas,
                      real* const phi);
void preq_omega_ps(const real* const p, const real* const vgrad_p, const real* const divdp, real* const omega_p);
// sphere_operators.hpp:9-16
void gradient_sphere(const real* const s, const TestData& data, int ielem, real* const ds);
void divergence_sphere(const real* const v, const TestData& data, int ielem, real* const div);
void vorticity_sphere(const real* const v, const TestData& data, int ielem, real* const vort);
real compute_norm(const real* const field, int length);
void print_results_2norm(const TestData& data);
void dump_results_to_file(const TestData& data);

// Element arrays resident on one GPU.  Not copyable; all calls from one thread.
class DeviceSession {
 public:
  // uploads all 16 arrays of `data` (num_elems elements) to HIP device `device`
  explicit DeviceSession(const TestData& data, int num_elems, int device = 0);
  // the same for the slab [first_elem, first_elem + num_elems) of data's arrays: one session
  // per GPU when the element range is sharded (elements are independent: no exchange).
  // run()/state_norms() then take Control::nets/nete relative to the slab.
  DeviceSession(const TestData& data, int first_elem, int num_elems, int device);
  ~DeviceSession();
  DeviceSession(const DeviceSession&) = delete;
  DeviceSession& operator=(const DeviceSession&) = delete;

  // host -> device copy of all 16 arrays again (after the host changed them)
  void upload(const TestData& data);
  // Vertical coordinate of the following run() calls.  Default rsplit = 1: vertically
  // Lagrangian, the reference's path (its pointers_only Control has no rsplit).  rsplit = 0:
  // Eulerian (eta_dot_dpdn and vertical advection computed; routine_extracted.F90:224-262),
  // hybi = nlev+1 interface coefficients, copied.  Parity of that branch is unpinned.
  void set_vertical_coordinate(int rsplit, const real* hybi);
  // one compute_and_apply_rhs with data's current Control/Constants/HVCoord/Derivative;
  // asynchronous, ordered on the session's stream
  void run(const TestData& data);
  // `nsteps` runs as one hipGraph launch (caar_run_steps); rotate: TestData::update_time_levels
  // between them.  The caller still rotates `data` itself nsteps-1 times afterwards.
  void run_steps(const TestData& data, int nsteps, bool rotate);
  void sync();
  // copy the arrays the kernel mutates (all_arrays: every array) back into data's host arrays
  void download(TestData& data, bool all_arrays = false);
  // print_results_2norm's numbers for time level control.np1, computed on the device
  void state_norms(const TestData& data, real out[3]);
  // milliseconds for `reps` back-to-back runs after one warm-up run (host wall clock around enqueue + wait)
  float time_runs(const TestData& data, int reps);

 private:
  CaarContext* ctx_;
  int num_elems_;
  int first_elem_;
  int rsplit_;
  real hybi_[nlev + 1];
};

}  // namespace Homme
#endif
