"""Where a GpuVecEnv.step (numpy boundary) spends its time: cProfile over 500 steps.   python tools/vecenv_profile.py [--vehicle hexa_arm]"""
import argparse
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--vehicle", default="hexa_arm")
a = ap.parse_args()
import rl_aerial_manipulator_amd as amd

env = amd.GpuVecEnv(num_envs=a.envs, vehicle=a.vehicle, seed=0)
env.reset()
rng = np.random.RandomState(0)
act = rng.normal(0, 0.1, (64, a.envs, env.action_space.shape[0])).astype(np.float32)
act[..., 0] += 1.0
act = np.clip(act, env.action_space.low, env.action_space.high)
for t in range(200):
    env.step(act[t % 64])
pr = cProfile.Profile()
pr.enable()
for t in range(500):
    env.step(act[t % 64])
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
