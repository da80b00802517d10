"""PCIe-inclusive rate of the SB3-shaped adapter (numpy in / numpy out): GpuVecEnv.step with host-resident actions,
for INTEGRATION.md section 5.  Never the headline value (that one is measured with inputs resident in HBM).

    python tools/vecenv_rate.py [--envs 4096] [--steps 2000] [--vehicle quad]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--vehicle", default="quad")
    a = ap.parse_args()
    import rl_aerial_manipulator_amd as amd
    env = amd.GpuVecEnv(num_envs=a.envs, vehicle=a.vehicle, seed=0)
    env.reset()
    rng = np.random.RandomState(0)
    act = rng.normal(0, 0.1, (64, a.envs, env.action_space.shape[0])).astype(np.float32)
    act[..., 0] += 1.0
    act = np.clip(act, env.action_space.low, env.action_space.high)
    for t in range(100):
        env.step(act[t % 64])
    t0 = time.perf_counter()
    events = 0
    for t in range(a.steps):
        _, _, done, infos = env.step(act[t % 64])
        events += int(done.sum())
    dt = time.perf_counter() - t0
    print(json.dumps({"adapter": "GpuVecEnv (numpy boundary, info dicts built for env events)", "envs": a.envs, "vehicle": a.vehicle,
                      "env_steps_per_s": a.envs * a.steps / dt, "us_per_vec_step": dt / a.steps * 1e6, "episodes_ended": events}))
