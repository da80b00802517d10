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
ap.add_argument("--kernel", default="auto")
ap.add_argument("--calibrate", action="store_true", help="also run a 256 MiB device copy (known byte count)")
a = ap.parse_args()
import torch

import rl_aerial_manipulator_amd as amd

env = amd.GpuWaypointEnv(a.envs, vehicle=a.vehicle, seed=0, block_size=a.block_size, kernel=a.kernel)
env.reset()
g = torch.Generator(device="cuda").manual_seed(1)
ring = torch.randn(16, a.envs, env.act_dim, device="cuda", generator=g) * 0.1
ring[..., 0] += 1.0
ring[..., 4:] *= 3.0
ring = ring.clamp(min=-1, max=2).contiguous()
for t in range(a.steps):
    env.step(ring[t % 16])
if a.calibrate:
    # a copy of known size done by an ELEMENTWISE KERNEL (16 B per lane): a plain Tensor.copy_ may go through the SDMA engine or a blit
    # kernel of another name and then the counter pass holds no dispatch to calibrate on
    src = torch.empty(256 * 1024 * 1024 // 4, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    for _ in range(3):
        torch.mul(src, 1.0, out=dst)
torch.cuda.synchronize()
print("done", env.kernel_name, env.stats()["episodes"])
