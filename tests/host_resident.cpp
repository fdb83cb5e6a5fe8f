// host_resident.cpp — test program (built and run by tests/test_host_driver_gpu.py on the GPU box, with
// CAAR_SHIM_RESIDENT=1): the resident mode of Homme::compute_and_apply_rhs(TestData&) (host/homme_caar.cpp).
//
//   (1) the reference driver's loop (main.cpp:113-121, with the rotation it has commented out) through the shim, then
//       sync_to_host: every array must equal, bit for bit, what the same loop gives through DeviceSession (the device
//       path that does not go through the shim's registry);
//   (2) before sync_to_host the host's copy of the np1 state is still the input (the documented staleness);
//   (3) sync_to_device: a change the host makes to its arrays reaches the device copy;
//   (4) a second set of arrays takes the device copy over and the first set's results are written back first;
//   (5) shim_stats counts the calls;
//   (6) (argv[2] = "misuse") sync_to_device on a stale host copy aborts.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "homme_caar.hpp"

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
typedef std::vector<std::vector<double>> Snap;
static Snap snapshot(const TestData& d, int ne) {
  Snap s(16);
  real* const* p = reinterpret_cast<real* const*>(&d.arrays);
  for (int i = 0; i < 16; ++i) s[i].assign(p[i], p[i] + array_len(i, ne));
  return s;
}
static int differing(const Snap& a, const Snap& b) {
  int n = 0;
  for (int i = 0; i < 16; ++i) n += std::memcmp(a[i].data(), b[i].data(), sizeof(double) * a[i].size()) != 0;
  return n;
}

int main(int argc, char** argv) {
  const int ne = argc > 1 ? std::atoi(argv[1]) : 23, calls = 3;
  num_elems = ne;
  int bad = 0;
  if (argc > 2 && std::strcmp(argv[2], "misuse") == 0) {
    // sync_to_device while the host copy is stale would undo the call: the shim must abort, not upload
    TestData m;
    m.init_data();
    compute_and_apply_rhs(m);
    sync_to_device(m);
    std::printf("sync_to_device on a stale host copy returned\nFAILED\n");
    return 1;
  }

  // the same loop through DeviceSession: what the shim must reproduce
  TestData want_d;
  want_d.init_data();
  {
    DeviceSession gpu(want_d, ne);
    for (int i = 0; i < calls; ++i) {
      gpu.run(want_d);
      if (i + 1 < calls) want_d.update_time_levels();
    }
    gpu.download(want_d, true);
  }
  const Snap want = snapshot(want_d, ne);

  TestData a;
  a.init_data();
  const Snap input = snapshot(a, ne);
  for (int i = 0; i < calls; ++i) {
    compute_and_apply_rhs(a);
    if (i + 1 < calls) a.update_time_levels();
  }
  ShimStats st = shim_stats();
  if (!st.resident) {
    std::printf("not in resident mode: run with CAAR_SHIM_RESIDENT=1\nFAILED\n");
    return 1;
  }
  if (st.calls != calls) ++bad;
  std::printf("(5) shim_stats: resident, %lld calls, %.6f s\n", st.calls, st.seconds);
  // (2)
  if (differing(snapshot(a, ne), input) != 0) {
    std::printf("(2) host arrays changed before sync_to_host\n");
    ++bad;
  } else {
    std::printf("(2) host arrays untouched before sync_to_host\n");
  }
  // (1)
  sync_to_host(a);
  int d = differing(snapshot(a, ne), want);
  std::printf("(1) after sync_to_host: %d arrays differ from the DeviceSession loop\n", d);
  bad += d;

  // (3) the host changes T at n0, tells the shim, calls once more; DeviceSession sees the same change through upload
  const size_t blk = size_t(nlev) * np * np;
  for (TestData* t : {&a, &want_d})
    for (size_t i = 0; i < blk; ++i) t->arrays.elem_state_T[(size_t(ne - 1) * timelevels + t->control.n0) * blk + i] += 0.25;
  sync_to_device(a);
  compute_and_apply_rhs(a);
  sync_to_host(a);
  {
    DeviceSession gpu(want_d, ne);
    gpu.run(want_d);
    gpu.download(want_d, true);
  }
  d = differing(snapshot(a, ne), snapshot(want_d, ne));
  std::printf("(3) after sync_to_device + one call: %d arrays differ\n", d);
  bad += d;

  // (4) another set of arrays: a's device copy is replaced; a call on `a` left un-synced is written back first
  compute_and_apply_rhs(a);  // doubles the accumulators once more; not synced
  {
    DeviceSession gpu(want_d, ne);
    gpu.run(want_d);
    gpu.download(want_d, true);
  }
  TestData b;
  b.init_data();
  compute_and_apply_rhs(b);  // takes the registry over: a's results go back to a's host arrays
  d = differing(snapshot(a, ne), snapshot(want_d, ne));
  std::printf("(4) first set written back when a second set took over: %d arrays differ\n", d);
  bad += d;
  sync_to_host(b);
  {
    TestData c;
    c.init_data();
    DeviceSession gpu(c, ne);
    gpu.run(c);
    gpu.download(c, true);
    d = differing(snapshot(b, ne), snapshot(c, ne));
    std::printf("    second set: %d arrays differ\n", d);
    bad += d;
    c.cleanup_data();
  }
  b.cleanup_data();  // release_host_mapping() inside
  a.cleanup_data();
  want_d.cleanup_data();
  std::printf(bad ? "FAILED\n" : "OK\n");
  return bad ? 1 : 0;
}
