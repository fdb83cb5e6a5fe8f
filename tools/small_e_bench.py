#!/usr/bin/env python3
"""Latency regime: time per call for element counts far below what fills the chip
(BASELINE configs[0] is 64 elements): plain stream launches, one hipGraph of 20 calls, and the 20 calls as ONE kernel
launch with rotating time levels (caar_launch_steps)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tinman_sandbox_amd as tsa  # noqa: E402

for E in (16, 64, 256, 1024, 4096):
    data = tsa.TestData().init_data(E, 4, 72, device="cuda")
    data.dvv_device()
    for _ in range(5):
        tsa.compute_and_apply_rhs(data)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        tsa.compute_and_apply_rhs(data)
    b.record()
    torch.cuda.synchronize()
    plain = a.elapsed_time(b) / 200
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            tsa.compute_and_apply_rhs(data)
    g.replay()
    torch.cuda.synchronize()
    a.record()
    for _ in range(10):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    graph = a.elapsed_time(b) / 200
    data.constants.eta_ave_w, data.control.dt2 = 0.0, 1e-6   # timing only: keeps hundreds of leap-frog steps finite
    tsa.compute_and_apply_rhs_steps(data, 20, True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(10):
        tsa.compute_and_apply_rhs_steps(data, 20, True)
    b.record()
    torch.cuda.synchronize()
    loop = a.elapsed_time(b) / 200
    print("E=%5d  stream launches %.2f us/call (%.2f M updates/s) | hipGraph of 20 %.2f us/call (%.2f M updates/s) | "
          "one launch of 20 calls %.2f us/call (%.2f M updates/s)" % (
              E, plain * 1e3, E / plain / 1e3, graph * 1e3, E / graph / 1e3, loop * 1e3, E / loop / 1e3), flush=True)
