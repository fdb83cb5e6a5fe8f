// caar_window_tuner.h — the adaptive cache window's state machine, free of HIP (plain C++17), so that it can be driven by a
// fake clock on the CPU (tests/window_tuner_sim.cpp) as well as by HIP events (caar_abi.hip).
//
// One WindowTunerState per array set.  step(backend) is called under the owner's mutex for every whole-range launch that the
// lock-free countdown did not let through; it returns the cache policy of THIS launch (1 = window, 0 = all streaming), may ask
// the backend to put a time stamp in front of the launch, and says how many of the FOLLOWING launches have nothing to do
// here (`idle_granted`: the owner lets that many pass without the mutex and reports them back through the same field).
//
// What is measured is the time from the start of one call to the start of the next (a call AND what it does to the
// neighbour that follows).  After kFirstProbe calls, whenever the current policy's smoothed call-to-call time drifts up by
// more than kDrift, every kReprobe calls while the policy is all-streaming and every kReprobeWindow while it is the window,
// the OTHER policy runs for kWarm + kMeas calls, then the current one for kWarm + kMeas, and the faster (medians of the
// measured calls; the window on ties) becomes the policy.  A probe that had to be discarded (a stamp failed, an interval
// was not positive) is retried kFirstProbe calls later.
//
// Backend (duck-typed): bool stamp(int slot)            put a time stamp in front of this launch into `slot`
//                       bool ready(int slot)            has the stamp in `slot` been reached?
//                       bool elapsed(int a, int b, float* ms)   time between the stamps of two slots (both reached), > 0
// Slots 0 .. kProbeEvents-1 are the probe's, kProbeEvents and kProbeEvents + 1 the passive sample's.
#pragma once

namespace caar {

struct WindowTunerState {
  static constexpr int kFirstProbe = 48, kWarm = 3, kMeas = 4, kHalf = kWarm + kMeas, kReprobe = 96, kReprobeWindow = 4096, kSampleEvery = 8;
  static constexpr int kProbeEvents = 2 * kHalf + 1;  // one in front of every probe call + one in front of the call after
  static constexpr int kSlots = kProbeEvents + 2;
  static constexpr double kDrift = 0.03, kTie = 0.003;

  long long calls = 0, since_decision = 0, probes = 0;
  long long idle_granted = 0;                  // in: launches that passed without step() since the last one; out: how many may
  int use_window = 1;                          // the policy in force
  double ms_window = 0.0, ms_streaming = 0.0;  // medians of the last probe (call-to-call times)
  double base_ms = 0.0, cur_ms = 0.0;          // the current policy's call-to-call time: at the decision / smoothed since
  // probe: 0 = idle; 1 .. 2 kHalf = that call of the probe is next; 2 kHalf + 1 = the closing stamp is next;
  // 2 kHalf + 2 = all stamps taken, waiting for the last one to be reached
  int probe_step = 0, probe_first = 0 /* policy of the probe's first half */;
  int sample = 0;  // 0 idle, 1 = first stamp taken at the previous call, 2 = both taken, waiting
  bool broken = false;

  void reset() { *this = WindowTunerState(); }

  static double median_of(float* v, int n) {
    for (int i = 1; i < n; ++i)
      for (int j = i; j > 0 && v[j] < v[j - 1]; --j) {
        const float t = v[j];
        v[j] = v[j - 1];
        v[j - 1] = t;
      }
    return n % 2 ? v[n / 2] : 0.5 * (v[n / 2 - 1] + v[n / 2]);
  }

  template <class Backend>
  int step(Backend& be) {
    // the launches that went by since the last step(), and this one
    calls += idle_granted + 1;
    since_decision += idle_granted + 1;
    idle_granted = 0;
    auto stamp = [&](int slot) {
      if (be.stamp(slot)) return true;
      broken = true;
      return false;
    };
    // the passive sample of the current policy: completed?
    if (sample == 2 && be.ready(kProbeEvents + 1)) {
      float ms;
      if (be.elapsed(kProbeEvents, kProbeEvents + 1, &ms) && probe_step == 0) cur_ms = cur_ms > 0.0 ? 0.75 * cur_ms + 0.25 * ms : ms;
      sample = 0;
    }
    // a probe whose stamps are all taken: decide once the last one has been reached
    if (probe_step == 2 * kHalf + 2 && (broken || be.ready(2 * kHalf))) {
      float m[2][kMeas];
      bool ok = !broken;
      for (int h = 0; h < 2 && ok; ++h)
        for (int i = 0; i < kMeas && ok; ++i) {
          const int c = h * kHalf + kWarm + i;  // 0-based probe call: from its start to the next call's start
          ok = be.elapsed(c, c + 1, &m[h][i]);
        }
      if (ok) {
        const double first = median_of(m[0], kMeas), second = median_of(m[1], kMeas);
        ms_window = probe_first ? first : second;
        ms_streaming = probe_first ? second : first;
        use_window = ms_window <= ms_streaming * (1.0 + kTie) ? 1 : 0;
        base_ms = cur_ms = use_window ? ms_window : ms_streaming;
        ++probes;
      }
      probe_step = 0;
      broken = false;
      since_decision = 0;  // (also after a discarded probe: the first-probe test below re-arms on it)
      sample = 0;
    }
    if (probe_step == 0) {
      const bool first = probes == 0 && since_decision >= kFirstProbe;
      const bool drift = probes > 0 && base_ms > 0.0 && cur_ms > base_ms * (1.0 + kDrift) && since_decision > 24;
      // (while the window is the policy a re-probe costs seven all-streaming calls, so it is rare: it only guards against a
      // first probe that was taken during a fresh process's ramp-up, which favours whichever policy ran second)
      const bool again = probes > 0 && since_decision >= (use_window ? kReprobeWindow : kReprobe);
      if (first || drift || again) {
        probe_step = 1;
        sample = 0;
        probe_first = use_window ? 0 : 1;  // the OTHER policy first, the current one second
      }
    }
    int policy = use_window;
    if (probe_step >= 1 && probe_step <= 2 * kHalf) {
      const int s = probe_step++;  // 1 .. 2 kHalf
      (void)stamp(s - 1);
      policy = (s - 1) / kHalf == 0 ? probe_first : 1 - probe_first;
    } else if (probe_step == 2 * kHalf + 1) {  // the call after the probe: its start closes the last measured interval
      (void)stamp(2 * kHalf);
      probe_step = 2 * kHalf + 2;
    } else if (probe_step == 0) {
      if (sample == 1) {
        sample = stamp(kProbeEvents + 1) ? 2 : 0;
      } else if (sample == 0 && calls % kSampleEvery == 0) {
        sample = stamp(kProbeEvents) ? 1 : 0;
      }
    }
    // How many of the following launches have nothing to do here: none while a probe's calls run or a sample's second
    // stamp is due; else up to the next sample slot or the first probe, whichever comes first.  (Also while a sample or a
    // finished probe only WAITS for its last stamp to be reached: a host that enqueues far ahead of the GPU would otherwise
    // come here on every call until the GPU has caught up; completion is looked at again at the next slot.)
    long long grant = 0;
    if ((probe_step == 0 && sample != 1) || probe_step == 2 * kHalf + 2) {
      grant = kSampleEvery - 1 - calls % kSampleEvery;
      if (probe_step == 0 && probes == 0) {
        const long long to_first = kFirstProbe - 1 - since_decision;
        if (to_first < grant) grant = to_first < 0 ? 0 : to_first;
      }
    }
    idle_granted = grant;
    return policy;
  }
};

}  // namespace caar
