// host_reentrancy.cpp — test program (built and run by tests/test_host_driver_gpu.py on the GPU box).
//
// (1) SURVEY 8b "Threading": the reference is re-entrant on disjoint [nets, nete) with separate Control copies
//     (HOMME's horizontal OpenMP).  N host threads call Homme::compute_and_apply_rhs on disjoint slabs of ONE
//     set of arrays, each through its own TestData copy (shallow: same array pointers, own Control); the result
//     must equal the single-thread call over the whole range bit for bit.
// (2) The reference-signature operator functions of the shim (sphere_operators.hpp:9-16,
//     compute_and_apply_rhs.hpp:11-17) against the oracle (test infrastructure: this program links it).
// (3) (argv[3] = elements, default 0 = skip) The adaptive cache window is off the launch path: 8 host
//     threads x 200 caar_launch calls on disjoint eighths of one device-resident array set, with the adaptive window on and
//     with caar_set_adaptive_window(0): the tuner's mutex is never taken by a sub-range launch (caar_adaptive_window_lock_count
//     does not move), the results are bit-identical, and the wall time is printed for both (the caller asserts 3 %).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <chrono>

#include "caar.h"
#include "caar_tuning.h"
#include "homme_caar.hpp"
extern "C" {
#include "caar_oracle.h"
}

using namespace Homme;
namespace Homme { int num_elems = 0; }

static size_t array_len(int i, int ne) {
  const size_t pp = size_t(np) * np, blk = pp * nlev;
  switch (i) {
    case 0: case 1: return ne * pp * 4;
    case 2: case 3: case 4: case 5: case 9: return ne * pp;
    case 6: case 8: return ne * timelevels * blk;
    case 7: return ne * timelevels * blk * 2;
    case 10: return size_t(ne) * qsize_d * 2 * blk;
    case 11: return ne * (blk + pp);
    case 12: case 13: case 14: return ne * blk;
    default: return ne * blk * 2;
  }
}

static std::vector<std::vector<double>> snapshot(const TestData& d, int ne) {
  std::vector<std::vector<double>> s(16);
  real* const* p = reinterpret_cast<real* const*>(&d.arrays);
  for (int i = 0; i < 16; ++i) s[i].assign(p[i], p[i] + array_len(i, ne));
  return s;
}

int main(int argc, char** argv) {
  const int ne = argc > 1 ? std::atoi(argv[1]) : 37, nthreads = argc > 2 ? std::atoi(argv[2]) : 4;
  num_elems = ne;
  int bad = 0;

  // ---- (1) threads on disjoint element ranges --------------------------------------------------
  TestData ref;
  ref.init_data();
  compute_and_apply_rhs(ref);
  sync_to_host(ref);  // (resident mode, CAAR_SHIM_RESIDENT=1: the host arrays are stale until now; a no-op in mapped mode)
  const auto want = snapshot(ref, ne);
  ref.cleanup_data();

  TestData shared;
  shared.init_data();
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; ++t)
    th.emplace_back([&, t] {
      TestData mine = shared;  // same arrays, private Control (data_structures.hpp:58-69)
      mine.control.nets = int((long long)ne * t / nthreads);
      mine.control.nete = int((long long)ne * (t + 1) / nthreads);
      compute_and_apply_rhs(mine);
    });
  for (auto& t : th) t.join();
  sync_to_host(shared);
  const auto got = snapshot(shared, ne);
  for (int i = 0; i < 16; ++i)
    if (std::memcmp(got[i].data(), want[i].data(), sizeof(double) * want[i].size()) != 0) {
      std::printf("array %d differs between %d threads and one\n", i, nthreads);
      ++bad;
    }
  std::printf("threads: %d element ranges on %d threads %s the single call (%s mode)\n", nthreads, nthreads,
              bad ? "DIFFER from" : "== bitwise", shim_stats().resident ? "resident" : "mapped");

  // ---- (2) the operator functions -----------------------------------------------------------------
  const int pp = np * np;
  std::vector<double> s(pp), v(2 * pp), o1(2 * pp), o2(2 * pp);
  for (int i = 0; i < pp; ++i) s[i] = std::sin(0.37 * i) + 2.0;
  for (int i = 0; i < 2 * pp; ++i) v[i] = std::cos(0.11 * i) - 0.3;
  double worst = 0;
  auto cmp = [&](const std::vector<double>& a, const std::vector<double>& b, int n) {
    double m = 0, d = 0;
    for (int i = 0; i < n; ++i) {
      m = std::fmax(m, std::fabs(b[i]));
      d = std::fmax(d, std::fabs(a[i] - b[i]));
    }
    worst = std::fmax(worst, d / (m > 0 ? m : 1));
  };
  const double* Dvv = &shared.deriv.Dvv[0][0];
  for (int ie : {0, ne - 1}) {
    const Arrays& a = shared.arrays;
    gradient_sphere(s.data(), shared, ie, o1.data());
    oracle_gradient_sphere(np, s.data(), Dvv, a.elem_Dinv + size_t(ie) * pp * 4, shared.constants.rrearth, o2.data());
    cmp(o1, o2, 2 * pp);
    divergence_sphere(v.data(), shared, ie, o1.data());
    oracle_divergence_sphere(np, v.data(), Dvv, a.elem_Dinv + size_t(ie) * pp * 4, a.elem_metdet + size_t(ie) * pp,
                             a.elem_rmetdet + size_t(ie) * pp, shared.constants.rrearth, o2.data());
    cmp(o1, o2, pp);
    vorticity_sphere(v.data(), shared, ie, o1.data());
    oracle_vorticity_sphere(np, v.data(), Dvv, a.elem_D + size_t(ie) * pp * 4, a.elem_rmetdet + size_t(ie) * pp,
                            shared.constants.rrearth, o2.data());
    cmp(o1, o2, pp);
  }
  std::printf("operators: worst scaled error vs oracle %.3e\n", worst);
  if (!(worst <= 1e-13)) ++bad;

  const size_t blk = size_t(pp) * nlev;
  std::vector<double> phis(pp), Tv(blk), p(blk), dp(blk), vg(blk), dd(blk), r1(blk), r2(blk);
  for (int i = 0; i < pp; ++i) phis[i] = 100.0 + i;
  for (size_t i = 0; i < blk; ++i) {
    Tv[i] = 250.0 + std::sin(0.01 * i) * 30;
    dp[i] = 900.0 + std::cos(0.02 * i) * 300;
    p[i] = 1000.0 + 950.0 * (i / pp) + std::sin(0.03 * i);
    vg[i] = std::sin(0.05 * i) * 20;
    dd[i] = std::cos(0.07 * i) * 3;
  }
  preq_hydrostatic(phis.data(), Tv.data(), p.data(), dp.data(), 287.04, r1.data());
  oracle_preq_hydrostatic(np, nlev, phis.data(), Tv.data(), p.data(), dp.data(), 287.04, r2.data());
  const bool h_ok = std::memcmp(r1.data(), r2.data(), sizeof(double) * blk) == 0;
  preq_omega_ps(p.data(), vg.data(), dd.data(), r1.data());
  oracle_preq_omega_ps(np, nlev, p.data(), vg.data(), dd.data(), r2.data());
  const bool o_ok = std::memcmp(r1.data(), r2.data(), sizeof(double) * blk) == 0;
  std::printf("preq_hydrostatic %s, preq_omega_ps %s the oracle\n", h_ok ? "== bitwise" : "DIFFERS from", o_ok ? "== bitwise" : "DIFFERS from");
  if (!h_ok || !o_ok) ++bad;

  shared.cleanup_data();

  // ---- (3) the adaptive window and the launch path -----------------------------------------------------
  const int big = argc > 3 ? std::atoi(argv[3]) : 0;
  if (big > 0) {
    num_elems = big;
    TestData host;
    host.init_data();
    CaarDims d = {np, nlev, qsize_d, timelevels, big};
    CaarContext* ctx = nullptr;
    CaarArrays h, dev;
    std::memcpy(&h, &host.arrays, sizeof(h));
    if (caar_create(&ctx, &d, 0) || caar_upload(ctx, &h, 0, big) || caar_sync(ctx) || caar_device_arrays(ctx, &dev)) {
      std::printf("(3) set-up failed\nFAILED\n");
      return 1;
    }
    double* dvv_dev = nullptr;
    if (hipMalloc((void**)&dvv_dev, sizeof(double) * np * np) != hipSuccess ||
        hipMemcpy(dvv_dev, &host.deriv.Dvv[0][0], sizeof(double) * np * np, hipMemcpyHostToDevice) != hipSuccess) return 1;
    const int T = 8, CALLS = 200;
    std::vector<hipStream_t> streams(T);
    for (auto& st : streams)
      if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return 1;
    CaarParams base = {};
    base.n0 = 0; base.np1 = 1; base.nm1 = 2; base.qn0 = 0; base.dt2 = 1e-6; base.eta_ave_w = 0.0;  // (keeps 1 600 calls finite)
    base.rrearth = host.constants.rrearth; base.Rwater_vapor = host.constants.Rwater_vapor; base.Rgas = host.constants.Rgas;
    base.kappa = host.constants.kappa; base.ps0 = host.hvcoord.ps0; base.hyai0 = host.hvcoord.hyai[0];
    base.Dvv = &host.deriv.Dvv[0][0]; base.rsplit = 1;
    auto round = [&](int on, long long* locks) {
      caar_set_adaptive_window(on);
      (void)hipDeviceSynchronize();
      const long long l0 = caar_adaptive_window_lock_count();
      const auto t0 = std::chrono::steady_clock::now();
      std::vector<std::thread> ths;
      for (int t = 0; t < T; ++t)
        ths.emplace_back([&, t] {
          CaarParams p = base;
          p.nets = int((long long)big * t / T);
          p.nete = int((long long)big * (t + 1) / T);
          for (int i = 0; i < CALLS; ++i)
            if (caar_launch(&d, &dev, dvv_dev, &p, streams[t]) != CAAR_OK) std::abort();
        });
      for (auto& x : ths) x.join();
      (void)hipDeviceSynchronize();
      const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      *locks = caar_adaptive_window_lock_count() - l0;
      return s;
    };
    // a whole-range launch first, so that the set HAS a tuner entry the sub-range launches look up
    {
      CaarParams p = base;
      p.nets = 0;
      p.nete = big;
      for (int i = 0; i < 60; ++i)
        if (caar_launch(&d, &dev, dvv_dev, &p, streams[0]) != CAAR_OK) return 1;
      (void)hipDeviceSynchronize();
    }
    long long locks_on = 0, locks_off = 0, l;
    double best_on = 1e30, best_off = 1e30;
    round(1, &l);  // warm-up
    for (int rep = 0; rep < 5; ++rep) {  // alternating, best of five each
      best_off = std::fmin(best_off, round(0, &l));
      locks_off += l;
      best_on = std::fmin(best_on, round(1, &l));
      locks_on += l;
    }
    std::printf("(3) %d threads x %d sub-range launches on %d elements: adaptive on %.6f s, off %.6f s (ratio %.4f); tuner mutex taken %lld / %lld times\n",
                T, CALLS, big, best_on, best_off, best_on / best_off, locks_on, locks_off);
    if (locks_on != 0 || locks_off != 0) {
      std::printf("(3) a sub-range launch took the tuner's mutex\n");
      ++bad;
    }
    // whole-range launches from one thread: the mutex only where a sample or a probe step is due
    {
      CaarParams p = base;
      p.nets = 0;
      p.nete = big;
      caar_adaptive_window_reset();
      const long long l0 = caar_adaptive_window_lock_count();
      const int N = 400;
      for (int i = 0; i < N; ++i)
        if (caar_launch(&d, &dev, dvv_dev, &p, streams[0]) != CAAR_OK) return 1;
      (void)hipDeviceSynchronize();
      const long long took = caar_adaptive_window_lock_count() - l0;
      std::printf("(3) %d whole-range launches: tuner mutex taken %lld times\n", N, took);
      if (took <= 0 || took > N * 3 / 4) ++bad;  // (first call + 2-3 of every 8 + the 16 calls of the first probe)
    }
    for (auto& st : streams) (void)hipStreamDestroy(st);
    (void)hipFree(dvv_dev);
    caar_destroy(ctx);
    host.cleanup_data();
  }
  std::printf(bad ? "FAILED\n" : "OK\n");
  return bad ? 1 : 0;
}
