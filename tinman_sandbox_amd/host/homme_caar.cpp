// homme_caar.cpp — Homme::compute_and_apply_rhs and friends on top of include/caar.h.
#include "homme_caar.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "caar.h"

namespace Homme {

extern int num_elems;

namespace {

void check(int rc, const char* what) {
  if (rc != CAAR_OK) {
    // the reference's interface has no error channel (void return, P:15): fail loudly
    std::fprintf(stderr, "caar: %s failed: %d (%s)\n", what, rc, caar_strerror(rc));
    std::abort();
  }
}

CaarDims dims_for(int ne) {
  CaarDims d;
  d.np = np;
  d.nlev = nlev;
  d.qsize_d = qsize_d;
  d.timelevels = timelevels;
  d.num_elems = ne;
  return d;
}

CaarArrays host_arrays(const Arrays& a, int first_elem = 0);

CaarArrays host_arrays(const Arrays& a, int first_elem) {
  CaarArrays h;
  h.elem_D = a.elem_D;
  h.elem_Dinv = a.elem_Dinv;
  h.elem_fcor = a.elem_fcor;
  h.elem_spheremp = a.elem_spheremp;
  h.elem_metdet = a.elem_metdet;
  h.elem_rmetdet = a.elem_rmetdet;
  h.elem_state_dp3d = a.elem_state_dp3d;
  h.elem_state_v = a.elem_state_v;
  h.elem_state_T = a.elem_state_T;
  h.elem_state_phis = a.elem_state_phis;
  h.elem_state_Qdp = a.elem_state_Qdp;
  h.elem_derived_eta_dot_dpdn = a.elem_derived_eta_dot_dpdn;
  h.elem_derived_omega_p = a.elem_derived_omega_p;
  h.elem_derived_phi = a.elem_derived_phi;
  h.elem_derived_pecnd = a.elem_derived_pecnd;
  h.elem_derived_vn0 = a.elem_derived_vn0;
  if (first_elem > 0) {  // element-major arrays: a slab starts first_elem * (doubles per element) in
    const CaarDims one = dims_for(1);
    double** p = reinterpret_cast<double**>(&h);
    for (int i = 0; i < CAAR_NUM_ARRAYS; ++i) p[i] += (long long)first_elem * caar_array_len(&one, i);
  }
  return h;
}

CaarParams params_for(const TestData& d) {
  CaarParams p = {};  // also the padding: caar_run_steps keys its cached graph on the values
  p.nets = d.control.nets;
  p.nete = d.control.nete;
  p.n0 = d.control.n0;
  p.np1 = d.control.np1;
  p.nm1 = d.control.nm1;
  p.qn0 = d.control.qn0;
  p.dt2 = d.control.dt2;
  p.rrearth = d.constants.rrearth;
  p.eta_ave_w = d.constants.eta_ave_w;
  p.Rwater_vapor = d.constants.Rwater_vapor;
  p.Rgas = d.constants.Rgas;
  p.kappa = d.constants.kappa;
  p.ps0 = d.hvcoord.ps0;
  p.hyai0 = d.hvcoord.hyai[0];
  p.Dvv = &d.deriv.Dvv[0][0];
  // the reference's pointers_only Control has no rsplit: its path is the vertically
  // Lagrangian one (P:22-28: eta_dot_dpdn, T_vadv, v_vadv = 0)
  p.rsplit = 1;
  p.hybi = nullptr;
  p.hybi_dev = nullptr;
  return p;
}

}  // namespace

// ----------------------------------------------------------------- DeviceSession
DeviceSession::DeviceSession(const TestData& data, int ne, int device)
    : DeviceSession(data, 0, ne, device) {}

DeviceSession::DeviceSession(const TestData& data, int first_elem, int ne, int device)
    : DeviceSession(data, first_elem, ne, device, nullptr) {}

DeviceSession::DeviceSession(const TestData& data, int first_elem, int ne, int device, const CaarPlacement* placement)
    : ctx_(nullptr), num_elems_(ne), first_elem_(first_elem), rsplit_(1), hybi_() {
  const CaarDims d = dims_for(ne);
  if (!caar_supported(d.np, d.nlev)) check(CAAR_EUNSUPPORTED, "caar_supported(np, nlev)");
  check(caar_create_ex(&ctx_, &d, device, placement), "caar_create_ex");
  upload(data);
}

DeviceSession::~DeviceSession() { caar_destroy(ctx_); }

void DeviceSession::upload(const TestData& data) {
  const CaarArrays h = host_arrays(data.arrays, first_elem_);
  check(caar_upload(ctx_, &h, 0, num_elems_), "caar_upload");
}

void DeviceSession::set_vertical_coordinate(int rsplit, const real* hybi) {
  rsplit_ = rsplit;
  if (hybi) std::memcpy(hybi_, hybi, sizeof(hybi_));
}

void DeviceSession::run(const TestData& data) {
  CaarParams p = params_for(data);
  p.rsplit = rsplit_;
  p.hybi = hybi_;
  check(caar_run(ctx_, &p), "caar_run");
}

void DeviceSession::run_steps(const TestData& data, int nsteps, bool rotate) {
  CaarParams p = params_for(data);
  p.rsplit = rsplit_;
  p.hybi = hybi_;
  check(caar_run_steps(ctx_, &p, nsteps, rotate ? 1 : 0), "caar_run_steps");
}

void DeviceSession::sync() { check(caar_sync(ctx_), "caar_sync"); }

void DeviceSession::download(TestData& data, bool all_arrays) {
  const CaarArrays h = host_arrays(data.arrays, first_elem_);
  check(caar_download(ctx_, &h, 0, num_elems_, all_arrays ? 1 : 0), "caar_download");
  sync();
}

void DeviceSession::state_norms(const TestData& data, real out[3]) {
  check(caar_state_norms(ctx_, data.control.np1, data.control.nets, data.control.nete, out), "caar_state_norms");
}

float DeviceSession::time_runs(const TestData& data, int reps) {
  CaarParams p = params_for(data);
  p.rsplit = rsplit_;
  p.hybi = hybi_;
  float ms = 0.f;
  check(caar_time_runs(ctx_, &p, reps, &ms), "caar_time_runs");
  return ms;
}

// -------------------------------------------------- the reference's free functions
// P:15.  Host arrays in, host arrays out, synchronous.  The host's arrays are page-locked
// once (again only when TestData's pointers or the element count change) and the kernel
// reads and writes them in place over PCIe (caar_map_host / caar_run_mapped): every input
// byte crosses the link once, every output byte once, nothing is staged in HBM.  Still
// PCIe-bound by construction: hosts that step in a loop should hold a DeviceSession.
//
// Re-entrant like the reference (HOMME's horizontal OpenMP calls it from several host threads on
// disjoint [nets, nete) with their own Control copies): the page-lock registry below is guarded by a mutex and
// every call holds a reference to the mapping it runs on, so a caller that triggers a re-mapping (other
// arrays, other element count) cannot release it under a call still in flight; the launches themselves are
// serialised inside caar_run_mapped.  The registry is keyed on the 16 array pointers and the element count: a
// host that frees its arrays must call release_host_mapping() first (homme_data.cpp's cleanup_data does; the
// reference's own main.cpp frees only at exit) — page locks on freed memory would otherwise be reused should
// a later allocation land on the same addresses.
namespace {
struct Mapping {
  CaarHostMapping* m = nullptr;
  CaarArrays key;
  int ne = -1;
  ~Mapping() {
    if (m) (void)caar_unmap_host(m);  // the arrays may already be freed: not an error here
  }
};
std::mutex& registry_mutex() {
  static std::mutex* mu = new std::mutex();  // never destroyed: no HIP calls from static destructors at exit
  return *mu;
}
std::shared_ptr<Mapping>& registry() {
  static std::shared_ptr<Mapping>* r = new std::shared_ptr<Mapping>();
  return *r;
}
std::shared_ptr<Mapping> acquire_mapping(const CaarArrays& h, int ne) {
  std::lock_guard<std::mutex> g(registry_mutex());
  std::shared_ptr<Mapping>& cur = registry();
  if (!cur || cur->ne != ne || std::memcmp(&cur->key, &h, sizeof(h)) != 0) {
    cur.reset();  // unmapped now, or when the last call still running on it returns
    const CaarDims d = dims_for(ne);
    if (!caar_supported(d.np, d.nlev)) check(CAAR_EUNSUPPORTED, "caar_supported(np, nlev)");
    std::shared_ptr<Mapping> m = std::make_shared<Mapping>();
    check(caar_map_host(&m->m, &d, &h, 0), "caar_map_host");
    m->key = h;
    m->ne = ne;
    cur = m;
  }
  return cur;
}
}  // namespace

void release_host_mapping() {
  std::lock_guard<std::mutex> g(registry_mutex());
  registry().reset();
}

void compute_and_apply_rhs(TestData& data) {
  const int ne = data.control.nete > num_elems ? data.control.nete : num_elems;
  const std::shared_ptr<Mapping> map = acquire_mapping(host_arrays(data.arrays), ne);
  const CaarParams p = params_for(data);
  check(caar_run_mapped(map->m, &p), "caar_run_mapped");
}

// ---------------------------------------------------------- the reference's operator functions
// sphere_operators.hpp:9-16 and compute_and_apply_rhs.hpp:11-17 declare them next to the main routine: one np x np
// field of one element (one column set of one element) in host memory in, one out.  Each call is a few hundred
// bytes: upload, one launch of the device operator, download (caar_sphere_operator_host / caar_preq_*_host) —
// latency, not bandwidth; hosts that care batch over levels and elements through the device-pointer entry points.
// Thread-safe (the library keeps one device scratch area behind a mutex).
namespace {
void run_operator(int which, const real* in, const TestData& data, int ielem, real* out) {
  const CaarDims d = dims_for(ielem + 1);
  const CaarArrays h = host_arrays(data.arrays);
  check(caar_sphere_operator_host(&d, &h, &data.deriv.Dvv[0][0], which, ielem, in, out, data.constants.rrearth),
        "caar_sphere_operator_host");
}
}  // namespace

// sphere_operators.hpp:9-10 (S:9-48)
void gradient_sphere(const real* const s, const TestData& data, int ielem, real* const ds) {
  run_operator(0, s, data, ielem, ds);
}
// sphere_operators.hpp:12-13 (S:50-89)
void divergence_sphere(const real* const v, const TestData& data, int ielem, real* const div) {
  run_operator(1, v, data, ielem, div);
}
// sphere_operators.hpp:15-16 (S:91-129)
void vorticity_sphere(const real* const v, const TestData& data, int ielem, real* const vort) {
  run_operator(2, v, data, ielem, vort);
}

// compute_and_apply_rhs.hpp:11-13 (P:280-312)
void preq_hydrostatic(const real* const phis, const real* const T_v, const real* const p, const real* dp, real Rgas,
                      real* const phi) {
  const CaarDims d = dims_for(1);
  check(caar_preq_hydrostatic_host(&d, phis, T_v, p, dp, Rgas, phi), "caar_preq_hydrostatic_host");
}

// compute_and_apply_rhs.hpp:15-17 (P:314-352)
void preq_omega_ps(const real* const p, const real* const vgrad_p, const real* const divdp, real* const omega_p) {
  const CaarDims d = dims_for(1);
  check(caar_preq_omega_ps_host(&d, p, vgrad_p, divdp, omega_p), "caar_preq_omega_ps_host");
}

// P:353-370
real compute_norm(const real* const field, int length) {
  real norm = 0, c = 0;
  for (int i = 0; i < length; ++i) {
    const real y = field[i] * field[i] - c;
    const real t = norm + y;
    c = (t - norm) - y;
    norm = t;
  }
  return std::sqrt(norm);
}

// P:372-399 (host arrays; the device-resident equivalent is DeviceSession::state_norms)
void print_results_2norm(const TestData& data) {
  const std::size_t blk = std::size_t(nlev) * np * np;
  real vn = 0, tn = 0, dn = 0;
  for (int ie = data.control.nets; ie < data.control.nete; ++ie) {
    const std::size_t slab = std::size_t(ie) * timelevels + data.control.np1;
    vn += std::pow(compute_norm(data.arrays.elem_state_v + slab * blk * 2, int(blk * 2)), 2);
    tn += std::pow(compute_norm(data.arrays.elem_state_T + slab * blk, int(blk)), 2);
    dn += std::pow(compute_norm(data.arrays.elem_state_dp3d + slab * blk, int(blk)), 2);
  }
  std::cout << "   ---> Norms:\n"
            << "          ||v||_2  = " << std::setprecision(17) << std::sqrt(vn) << "\n"
            << "          ||T||_2  = " << std::setprecision(17) << std::sqrt(tn) << "\n"
            << "          ||dp||_2 = " << std::setprecision(17) << std::sqrt(dn) << "\n";
}

// P:401-487: one text file per field, "[ie, ilev]" header then np rows of np values at
// 6 significant digits.
void dump_results_to_file(const TestData& data) {
  struct Out {
    const char* name;
    const real* base;
    int comps, comp;
  };
  const Out outs[] = {{"elem_state_vx.txt", data.arrays.elem_state_v, 2, 0},
                      {"elem_state_vy.txt", data.arrays.elem_state_v, 2, 1},
                      {"elem_state_t.txt", data.arrays.elem_state_T, 1, 0},
                      {"elem_state_dp3d.txt", data.arrays.elem_state_dp3d, 1, 0}};
  const std::size_t pp = std::size_t(np) * np, blk = pp * nlev;
  for (const Out& o : outs) {
    std::ofstream f(o.name);
    if (!f.is_open()) {
      std::cout << "Error! Cannot open '" << o.name << "'.\n";
      std::abort();
    }
    f.precision(6);
    for (int ie = data.control.nets; ie < data.control.nete; ++ie) {
      const real* lev0 = o.base + (std::size_t(ie) * timelevels + data.control.np1) * blk * o.comps;
      for (int k = 0; k < nlev; ++k) {
        f << "[" << ie << ", " << k << "]\n";
        for (int i = 0; i < np; ++i) {
          for (int j = 0; j < np; ++j) f << " " << lev0[(k * pp + i * np + j) * o.comps + o.comp];
          f << "\n";
        }
      }
    }
  }
}

}  // namespace Homme
