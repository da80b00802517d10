#!/usr/bin/env python3
"""Fixed cost of one timed bench window (K steps in one hipGraph, bracketed by synchronize): where the ~25 us beyond K x kernel time go.
  python tools/window_overhead.py [--steps 20]"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
import torch

import rl_aerial_manipulator_amd as amd

env = amd.GpuWaypointEnv(4096, vehicle="hexa_arm", seed=0)
env.reset()
gen = torch.Generator(device="cuda").manual_seed(1)
ring = torch.randn(64, 4096, env.act_dim, device="cuda", generator=gen) * 0.1
ring[..., 0] += 1.0
ring = ring.clamp(min=-1, max=2).contiguous()
for t in range(64):
    env.step(ring[t])
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for t in range(a.steps):
        env.step(ring[t])
for _ in range(200):
    g.replay()
torch.cuda.synchronize()


def window(wait):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    g.replay()
    t1 = time.perf_counter()
    e1.record()
    t2 = time.perf_counter()
    if wait == "spin":
        while not e1.query():
            pass
    elif wait == "stream":
        torch.cuda.current_stream().synchronize()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    return (t3 - t0) * 1e6, e0.elapsed_time(e1) * 1e3, (t1 - t0) * 1e6, (t2 - t1) * 1e6


for wait in ("device", "stream", "spin"):
    rows = [window(wait) for _ in range(200)][50:]
    med = [statistics.median(r[k] for r in rows) for k in range(4)]
    print(f"wait={wait:6s}: wall {med[0]:7.1f} us, device (events) {med[1]:7.1f} us, replay() call {med[2]:5.1f} us, event record {med[3]:4.1f} us  -> {a.steps * 4096 / med[0] * 1e6:.4g} env-steps/s")
