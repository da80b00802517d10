"""Closed-loop rollout rate: policy inference + Gaussian sampling + env step, everything on the GPU (what `PPO.collect_rollouts`
and `evaluate_policy` do per step).  Compares the one-launch policy forward (`amenv_policy_forward`) with the torch modules.

    python tools/rollout_rate.py [--envs 4096] [--steps 2000] [--vehicle quad]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, nargs="+", default=[4096, 32768])
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--vehicle", default="quad")
    a = ap.parse_args()
    import torch
    import rl_aerial_manipulator_amd as amd
    out = {}
    for n in a.envs:
        env = amd.GpuWaypointEnv(n, vehicle=a.vehicle, seed=0)
        pol = amd.ActorCritic(env.obs_dim, env.act_dim).to(env.device).flatten_()
        for mode in ("fused", "torch"):
            obs = env.reset()
            if mode == "torch":
                pol.fused_ok = lambda o: False                        # instance override -> the torch modules
            with torch.no_grad():
                for _ in range(50):
                    obs, _, _, _ = env.step(pol.predict(obs))
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(a.steps):
                    obs, _, _, _ = env.step(torch.minimum(torch.maximum(pol.actor(obs), pol.action_low), pol.action_high))
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    for _ in range(16):
                        obs, _, _, _ = env.step(torch.minimum(torch.maximum(pol.actor(obs), pol.action_low), pol.action_high))
                torch.cuda.synchronize(); t1 = time.perf_counter()
                for _ in range(a.steps // 16):
                    graph.replay()
                torch.cuda.synchronize(); dg = time.perf_counter() - t1
            if mode == "torch":
                del pol.fused_ok
            out[f"{n}_{mode}"] = {"eager_us_per_step": dt / a.steps * 1e6, "graph_us_per_step": dg / (a.steps // 16 * 16) * 1e6,
                                  "graph_env_steps_per_s": n * (a.steps // 16 * 16) / dg}
        env.close()
    print(json.dumps({"vehicle": a.vehicle, "loop": "obs -> policy mean -> clip -> env.step", "results": out}))
