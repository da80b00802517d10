#!/usr/bin/env python3
"""In-kernel phase timing of the step kernel from a -DAMENV_STAMPS diagnostic build (GPU box).
  AMENV_LIB=<stamps.so> python tools/stamp_profile.py --envs 4096
Prints the median over launches and waves of each phase's share (shader-clock cycles, s_memtime)."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--vehicle", default="hexa")
ap.add_argument("--launches", type=int, default=100)
ap.add_argument("--hover", action="store_true")
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--kernel", default="auto")
a = ap.parse_args()
import torch

import rl_aerial_manipulator_amd as amd

env = amd.GpuWaypointEnv(a.envs, vehicle=a.vehicle, seed=a.seed, kernel=a.kernel)
env.reset()
lib = C.CDLL(amd._lib.LIB_PATH)
lib.amenv_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
g = torch.Generator(device="cuda").manual_seed(1)
ring = torch.randn(16, a.envs, env.act_dim, device="cuda", generator=g) * (0.0 if a.hover else 0.1)
ring[..., 0] += 1.0
ring[..., 4:] *= 3.0
ring = ring.clamp(min=-1, max=2).contiguous()
names = ["entry->loads issued", "loads issued->landed", "compute (mixer+RK4+task+obs)", "state/output stores issued", "LDS stage+barrier+obs flush",
         "stats", "drain stores (vmcnt 0)"]
if "arm2w" in env.kernel_name:   # two-wave arm kernel: stamps of the MAIN wave of each tile
    names = ["entry->loads issued", "loads issued->landed", "RK4 (with helper) + task step", "stats + state/output stores issued",
             "final barrier (helper's obs rows landed)", "reset rows (rare)", "drain stores (vmcnt 0)"]
if "team" in env.kernel_name:    # lane-team kernel: one wave = 4 envs; stamps of the first 64 waves
    names = ["entry->loads issued", "loads issued->landed", "-", "mixer + RK4 + forward kinematics + task step + episode end / reset",
             "Monitor totals", "stores issued", "drain stores (vmcnt 0)"]
rows, raws = [], []
nw = min(64, (a.envs + 63) // 64) if "team" not in env.kernel_name else min(64, (a.envs + 3) // 4)
for t in range(a.launches):
    env.step(ring[t % 16])
    buf = np.zeros((64, 8), np.uint64)
    lib.amenv_debug_stamps(env._h, buf.ctypes.data_as(C.c_void_p))
    raws.append(buf[:nw].astype(np.int64))
    d = np.diff(buf[:nw].astype(np.int64), axis=1)
    rows.append(d)
d = np.concatenate(rows[10:], 0)
med = np.median(d, 0)
tot = np.median((np.concatenate([r for r in rows[10:]], 0)).sum(1))
for n_, m in zip(names, med):
    print(f"{n_:34s} {m:8.0f} cycles  {100*m/med.sum():5.1f}%")
print(f"{'wave lifetime (entry->drained)':34s} {tot:8.0f} cycles")
# the kernel is as slow as its slowest wave: per launch, the wave with the longest lifetime
slow = np.stack([r[np.argmax(r.sum(1))] for r in rows[10:]], 0)
ms = np.median(slow, 0)
print("slowest wave of each launch (median over launches):")
for n_, m in zip(names, ms):
    print(f"  {n_:32s} {m:8.0f} cycles")
print(f"  {'lifetime':32s} {np.median(slow.sum(1)):8.0f} cycles   (fraction of launches whose slowest wave took the reset path: "
      f"{np.mean(slow[:, 2] > 1.15 * med[2]):.2f})")
# launch skew: first stamp of each wave relative to the earliest wave of the launch
R = np.stack(raws[10:], 0)                       # [launch, wave, stamp]
D = np.diff(R, axis=2)
print("median compute cycles by wave index (8 per row):")
mc = np.median(D[:, :, 2], 0)
for r0 in range(0, nw, 8):
    print("  " + " ".join(f"{x:6.0f}" for x in mc[r0:r0 + 8]))
print("median prologue (entry->loads issued) by wave index:")
mp_ = np.median(D[:, :, 0], 0)
for r0 in range(0, nw, 8):
    print("  " + " ".join(f"{x:6.0f}" for x in mp_[r0:r0 + 8]))
start = R[:, :, 0] - R[:, :, 0].min(1, keepdims=True)
end = R[:, :, 7] - R[:, :, 0].min(1, keepdims=True)
print(f"launch skew: last wave starts {np.median(start.max(1)):.0f} cycles after the first; kernel span first-entry -> last-drain {np.median(end.max(1)):.0f} cycles")
print("median start offset by wave index:")
ms_ = np.median(start, 0)
for r0 in range(0, nw, 8):
    print("  " + " ".join(f"{x:6.0f}" for x in ms_[r0:r0 + 8]))
