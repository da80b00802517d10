#!/usr/bin/env python3
"""Timeline of the three waves of step_kernel_pw (diagnostic -DAMENV_STAMPS build): where launches with an episode end spend their time.
  AMENV_LIB=tools/micro/libamenv_stamps.so python tools/stamp_pw.py --vehicle hexa
Stamps (cycles after the launch's earliest wave entry), median over launches, for tiles with / without an episode end in that launch."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--vehicle", default="hexa")
ap.add_argument("--launches", type=int, default=400)
a = ap.parse_args()
import torch

import rl_aerial_manipulator_amd as amd

env = amd.GpuWaypointEnv(a.envs, vehicle=a.vehicle, seed=0)
env.reset()
assert "step_kernel_pw" in env.kernel_name, env.kernel_name
lib = C.CDLL(amd._lib.LIB_PATH)
lib.amenv_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
g = torch.Generator(device="cuda").manual_seed(1)
ring = (torch.randn(16, a.envs, env.act_dim, device="cuda", generator=g) * 0.1)
ring[..., 0] += 1.0
ring = ring.clamp(min=-1, max=2).contiguous()
for t in range(2000):
    env.step(ring[t % 16])
recs = []
for t in range(a.launches):
    env.step(ring[t % 16])
    buf = np.zeros((64, 8), np.uint64)
    lib.amenv_debug_stamps(env._h, buf.ctypes.data_as(C.c_void_p))
    recs.append(buf[:64].astype(np.int64).reshape(16, 4, 8))
R = np.stack(recs, 0)                                  # [launch, tile, role, slot]
t0 = R[:, :, 0:1, 0:1]                                # entry of the tile's main wave
ended = R[:, :, 0, 7] != 0                              # [launch, tile]
rel = R[:, :, :, :7] - t0
names = {0: ["entry", "loads landed", "step_lane done", "flags written", "barrier passed", "stores issued", "drained"],
         1: ["entry", "words done", "reset state + obs done", "(same)", "barrier passed", "stores issued", "drained"],
         2: ["entry", "loads landed", "dynamics done", "obs rows flushed", "barrier passed", "cold stores issued", "drained"]}
print(f"{int(ended.sum())} tile-launches with an episode end, {int((~ended).sum())} without; launches with any end among the sampled tiles: {int(ended.any(1).sum())} of {a.launches}")
names[3] = ["entry", "-", "-", "-", "barrier passed", "cold stores / atomics issued", "drained"]
for role, rn in ((0, "main"), (1, "reset wave"), (2, "observation wave"), (3, "Monitor wave")):
    print(f"-- wave {role} ({rn}): median cycles since the entry of the tile's main wave   [no end | end in this tile]")
    for k, n_ in enumerate(names[role]):
        x0 = np.median(rel[:, :, role, k][~ended]); x1 = np.median(rel[:, :, role, k][ended]) if ended.any() else float('nan')
        print(f"   {n_:26s} {x0:8.0f} {x1:8.0f}")
for role in ():
    x = (R[:, :, role, 7] - R[:, :, 0, 0])[ended & (R[:, :, role, 7] != 0)]
    print(f"wave {role}: first instruction inside its episode-end branch at {np.median(x):.0f} (median)")
last = rel[:, :, :, 6].max(axis=2)                      # per tile: last wave drained
print(f"tile finished (last of its waves drained): median {np.median(last[~ended]):.0f} without end, {np.median(last[ended]):.0f} with end")
span = last.max(axis=1)
print(f"launch span over the sampled tiles: median {np.median(span[~ended.any(1)]):.0f} (no end among them) vs {np.median(span[ended.any(1)]):.0f} (some end)")
