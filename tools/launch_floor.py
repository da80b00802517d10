#!/usr/bin/env python3
"""Dependent-launch floor on the GPU box (diagnostic -DAMENV_STAMPS build): per-kernel period of a hipGraph chain of
(a) empty kernels, (b) one-load-one-store kernels, (c) the real step kernel, same grid.
  AMENV_LIB=<stamps.so> python tools/launch_floor.py --envs 4096"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
a = ap.parse_args()
import torch

import rl_aerial_manipulator_amd as amd

env = amd.GpuWaypointEnv(a.envs, vehicle="hexa", seed=0)
env.reset()
lib = C.CDLL(amd._lib.LIB_PATH)
lib.amenv_debug_noop.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
ring = torch.zeros(64, a.envs, 4, device="cuda"); ring[..., 0] = 1.0


def chain(fn, reps=64):
    for _ in range(64):
        fn(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for t in range(64):
            fn(t)
    for _ in range(4):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * 64)


s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for bs in (64, 256):
    print(f"noop  chain, block {bs:3d}: {chain(lambda t: lib.amenv_debug_noop(env._h, 0, bs, s())):.3f} us/kernel")
    print(f"touch chain, block {bs:3d}: {chain(lambda t: lib.amenv_debug_noop(env._h, 1, bs, s())):.3f} us/kernel")
for grid, bs in ((256, 64), (1024, 64), (4096, 64), (256, 256), (1024, 256)):   # explicit grids: how the floor depends on the number of workgroups
    print(f"noop  chain, grid {grid:4d} x {bs:3d}: {chain(lambda t: lib.amenv_debug_noop(env._h, grid, bs, s())):.3f} us/kernel")
print(f"step  chain            : {chain(lambda t: env.step(ring[t])):.3f} us/kernel")
# eager (no graph) back-to-back
for name, fn in (("noop", lambda t: lib.amenv_debug_noop(env._h, 0, 64, s())), ("step", lambda t: env.step(ring[t % 64]))):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(4096):
        fn(t)
    e1.record(); torch.cuda.synchronize()
    print(f"{name} eager            : {e0.elapsed_time(e1) * 1e3 / 4096:.3f} us/kernel")
