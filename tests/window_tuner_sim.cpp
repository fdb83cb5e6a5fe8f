// window_tuner_sim.cpp — CPU test of the adaptive cache window's state machine (csrc/caar_window_tuner.h, no HIP) under a
// fake clock.  Built and run by tests/test_host.py::test_window_tuner_state_machine_under_a_fake_clock.
//
// The simulated device runs launches back to back; a launch costs cost(policy, call) ms; a stamp in front of launch i reads
// the start time of launch i and is "reached" once the device has got there — the host may run `lag` launches ahead.  The
// driver loop below is the owner's (caar_abi.hip adaptive_window_policy): a countdown lets `idle_granted` launches pass
// without step().
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include "caar_window_tuner.h"

using caar::WindowTunerState;

struct Sim {
  std::function<double(int policy, long long call)> cost;
  long long lag = 0;          // launches the host is ahead of the device
  long long fail_stamp_at = -1;  // the n-th stamp() fails (a discarded probe)
  std::vector<double> start{0.0};  // start[i] = start time of launch i
  long long stamp_launch[WindowTunerState::kSlots];
  long long stamps = 0, steps = 0;
  long long call = 0;
  Sim() { for (auto& x : stamp_launch) x = -1; }
  bool stamp(int slot) {
    if (stamps++ == fail_stamp_at) return false;
    stamp_launch[slot] = call;
    return true;
  }
  bool ready(int slot) const { return stamp_launch[slot] >= 0 && stamp_launch[slot] <= call - lag; }
  bool elapsed(int a, int b, float* ms) const {
    *ms = 0.f;
    if (stamp_launch[a] < 0 || stamp_launch[b] < 0) return false;
    *ms = float(start[stamp_launch[b]] - start[stamp_launch[a]]);
    return *ms > 0.f;
  }
  // run n launches; returns the policies they ran with
  std::vector<int> run(WindowTunerState& st, long long& idle, long long n) {
    std::vector<int> pol;
    for (long long i = 0; i < n; ++i, ++call) {
      int p;
      if (idle > 0) {
        --idle;
        p = st.use_window;
      } else {
        p = st.step(*this);
        idle = st.idle_granted;
        ++steps;
      }
      pol.push_back(p);
      start.push_back(start.back() + cost(p, call));
    }
    return pol;
  }
};

static int bad = 0;
#define CHECK(cond, ...)                         \
  do {                                           \
    if (!(cond)) {                               \
      std::printf("FAILED %s: ", #cond);         \
      std::printf(__VA_ARGS__);                  \
      std::printf("\n");                         \
      ++bad;                                     \
    }                                            \
  } while (0)

int main() {
  const int K = WindowTunerState::kFirstProbe, H = WindowTunerState::kHalf;
  {  // 1. a replaying host: the window is faster and stays
    Sim s;
    s.cost = [](int p, long long) { return p ? 0.300 : 0.350; };
    WindowTunerState st;
    long long idle = 0;
    const std::vector<int> pol = s.run(st, idle, 400);
    CHECK(st.probes == 1 && st.use_window == 1, "probes %lld use %d", st.probes, st.use_window);
    CHECK(st.ms_window > 0.299 && st.ms_window < 0.301 && st.ms_streaming > 0.349 && st.ms_streaming < 0.351, "%g %g", st.ms_window, st.ms_streaming);
    int streaming = 0;
    for (int i = 0; i < 400; ++i) streaming += pol[i] == 0;
    CHECK(streaming == H, "%d all-streaming calls, expected the probe's %d", streaming, H);
    for (int i = 0; i < K - 1; ++i) CHECK(pol[i] == 1, "call %d before the first probe", i);
    for (int i = K - 1; i < K - 1 + H; ++i) CHECK(pol[i] == 0, "probe call %d should try the other policy", i);
    CHECK(s.steps <= 400 * 3 / 8 + 2 * H + 4, "%lld of 400 launches took the slow path", s.steps);
    std::printf("1. replay: window kept, %lld of 400 launches on the slow path\n", s.steps);
  }
  {  // 2. a host whose neighbour evicts the cache: all-streaming wins, is re-probed every kReprobe calls, and stays
    Sim s;
    s.cost = [](int p, long long) { return p ? 0.352 : 0.347; };
    WindowTunerState st;
    long long idle = 0;
    s.run(st, idle, 500);
    CHECK(st.use_window == 0 && st.probes >= 4, "use %d probes %lld", st.use_window, st.probes);
    std::printf("2. evicting neighbour: all-streaming after %lld probes in 500 launches\n", st.probes);
  }
  {  // 3. the host's pattern changes: drift of the smoothed call-to-call time triggers a re-probe long before kReprobeWindow
    Sim s;
    s.cost = [](int p, long long c) { return c < 1000 ? (p ? 0.300 : 0.350) : (p ? 0.362 : 0.347); };
    WindowTunerState st;
    long long idle = 0;
    s.run(st, idle, 1000);
    CHECK(st.use_window == 1 && st.probes == 1, "before the change: use %d probes %lld", st.use_window, st.probes);
    const std::vector<int> pol = s.run(st, idle, 200);
    CHECK(st.use_window == 0, "after the change the policy should have flipped (probes %lld, cur %g base %g)", st.probes, st.cur_ms, st.base_ms);
    int first_streaming = -1;
    for (int i = 0; i < 200 && first_streaming < 0; ++i)
      if (pol[i] == 0) first_streaming = i;
    CHECK(first_streaming >= 0 && first_streaming < 120, "re-probe started %d launches after the change", first_streaming);
    std::printf("3. pattern change: re-probe %d launches after it, policy flipped\n", first_streaming);
  }
  {  // 4. a stamp fails inside the first probe: the probe is discarded and retried kFirstProbe launches later
    Sim s;
    s.cost = [](int p, long long) { return p ? 0.352 : 0.347; };
    s.fail_stamp_at = 12 + 5;  // (6 samples = 12 stamps before the probe; its 6th stamp fails)
    WindowTunerState st;
    long long idle = 0;
    s.run(st, idle, K + 2 * H + 8);
    CHECK(st.probes == 0 && st.probe_step == 0, "discarded: probes %lld step %d", st.probes, st.probe_step);
    s.run(st, idle, K + 2 * H + 16);
    CHECK(st.probes == 1 && st.use_window == 0, "retried: probes %lld use %d", st.probes, st.use_window);
    std::printf("4. discarded first probe retried: decided after %lld launches\n", s.call);
  }
  {  // 5. a host that enqueues 300 launches ahead of the device: decisions still arrive, and the slow path stays rare
    Sim s;
    s.cost = [](int p, long long) { return p ? 0.352 : 0.347; };
    s.lag = 300;
    WindowTunerState st;
    long long idle = 0;
    s.run(st, idle, 1200);
    CHECK(st.probes >= 1 && st.use_window == 0, "probes %lld use %d", st.probes, st.use_window);
    CHECK(s.steps <= 1200 * 3 / 8 + 64, "%lld of 1200 launches took the slow path", s.steps);
    std::printf("5. host 300 launches ahead: %lld probes decided, %lld of 1200 launches on the slow path\n", st.probes, s.steps);
  }
  {  // 6. ties keep the window
    Sim s;
    s.cost = [](int, long long) { return 0.300; };
    WindowTunerState st;
    long long idle = 0;
    s.run(st, idle, 200);
    CHECK(st.probes == 1 && st.use_window == 1, "tie: probes %lld use %d", st.probes, st.use_window);
  }
  std::printf(bad ? "FAILED\n" : "OK\n");
  return bad ? 1 : 0;
}
