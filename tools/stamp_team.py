#!/usr/bin/env python3
"""Lane-team step kernel (diagnostic -DAMENV_STAMPS build): phases of wavefronts with and without an episode end in the launch.
  AMENV_LIB=tools/micro/libamenv_stamps.so python tools/stamp_team.py"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--launches", type=int, default=1500)
a = ap.parse_args()
import torch

import rl_aerial_manipulator_amd as amd

env = amd.GpuWaypointEnv(a.envs, vehicle="hexa_arm", seed=0, kernel="team")
env.reset()
lib = C.CDLL(amd._lib.LIB_PATH)
lib.amenv_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
g = torch.Generator(device="cuda").manual_seed(1)
ring = torch.randn(16, a.envs, env.act_dim, device="cuda", generator=g) * 0.1
ring[..., 0] += 1.0
ring[..., 4:] *= 3.0
ring = ring.clamp(min=-1, max=2).contiguous()
for t in range(2000):
    env.step(ring[t % 16])
recs = []
for t in range(a.launches):
    env.step(ring[t % 16])
    buf = np.zeros((64, 8), np.uint64)
    lib.amenv_debug_stamps(env._h, buf.ctypes.data_as(C.c_void_p))
    recs.append(buf.astype(np.int64))
R = np.stack(recs, 0)                                   # [launch, wave (first 64 of 1024), slot]
ended = R[:, :, 3] != 0
t = R - R[:, :, 0:1]
print(f"{int(ended.sum())} wave-launches with an episode end, {int((~ended).sum())} without (64 of {a.envs // 4} waves sampled)")
for k, n_ in ((1, "loads issued"), (2, "loads landed"), (4, "advance done (RK4, task, episode end)"), (5, "barrier passed, reset values taken"), (6, "stores issued"), (7, "drained")):
    print(f"   {n_:40s} {np.median(t[:, :, k][~ended]):8.0f} {np.median(t[:, :, k][ended]):8.0f}")
