#!/usr/bin/env python3
"""What episode ends cost the step kernel: per-step time of a 64-step graph with (a) the bench's random actions in steady state (a few
envs of the batch end their episode in every launch) and (b) hover actions on a long horizon (no episode ends).
  python tools/reset_cost.py --vehicle hexa"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--vehicle", default="hexa")
ap.add_argument("--kernel", default="auto")
a = ap.parse_args()
import torch

import rl_aerial_manipulator_amd as amd


def rate(env, ring, reps=32):
    for t in range(64):
        env.step(ring[t % ring.shape[0]])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for t in range(64):
            env.step(ring[t % ring.shape[0]])
    for _ in range(64):
        g.replay()                                   # steady state: 4096 steps
    env.stats(reset=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * 64), env.stats()["episodes"] / (reps * 64)


gen = torch.Generator(device="cuda").manual_seed(1)
for name, kw, noise in (("random actions, default horizon", {}, 0.1), ("hover, horizon 10^6 steps", dict(max_episode_steps=1000000), 0.0),
                        ("hover, horizon 100 steps (time-limit ends only)", dict(max_episode_steps=100), 0.0)):
    env = amd.GpuWaypointEnv(a.envs, vehicle=a.vehicle, seed=0, kernel=a.kernel, **kw)
    env.reset()
    ring = torch.randn(16, a.envs, env.act_dim, device="cuda", generator=gen) * noise
    ring[..., 0] += 1.0
    ring = ring.clamp(min=-1, max=2).contiguous()
    us, eps = rate(env, ring)
    print(f"{a.vehicle:9s} {env.kernel_name[:34]:34s} {name:48s}: {us:.3f} us/step, {eps:.2f} episode ends per launch")
    env.close()
