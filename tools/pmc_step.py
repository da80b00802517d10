#!/usr/bin/env python3
"""Workload for rocprofv3 counter passes: N envs, a few hundred amenv_step launches, nothing else.
  rocprofv3 --pmc <counters> --output-format csv -d <dir> -- python3 tools/pmc_step.py --envs 4096 --steps 200"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--vehicle", default="hexa")
ap.add_argument("--block-size", type=int, default=0)
a = ap.parse_args()
import torch

import rl_aerial_manipulator_amd as amd

env = amd.GpuWaypointEnv(a.envs, vehicle=a.vehicle, seed=0, block_size=a.block_size)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
ring = torch.randn(16, a.envs, 4, device="cuda", generator=g) * 0.1
ring[..., 0] += 1.0
ring = ring.clamp(min=-1, max=2).contiguous()
for t in range(a.steps):
    env.step(ring[t % 16])
torch.cuda.synchronize()
print("done", env.kernel_name, env.stats()["episodes"])
