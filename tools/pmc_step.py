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
    # copies of known size (256 MiB) in the access pattern of the kernel under test: 16 B per lane (one-lane-per-env kernels) and 4 B per
    # lane (lane-team kernels), through the library's own calibration kernel so that the dispatch is found by name
    import ctypes as C
    src = torch.empty(256 * 1024 * 1024 // 4, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    for width in (16, 4):
        for _ in range(3):
            rc = env.lib.amenv_calibration_copy(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), src.numel() * 4, width, None)
            assert rc == 0
torch.cuda.synchronize()
print("done", env.kernel_name, env.stats()["episodes"])
