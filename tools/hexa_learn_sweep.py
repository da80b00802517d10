#!/usr/bin/env python3
"""Why does PPO with the reference's hyper-parameters not learn the hexacopter?  A controlled experiment.

Every run: amd.PPO from random weights (or from a behaviour-cloned PID policy), the reference's hyper-parameters (v2/rl_train.py:38-53), 256 envs x
512 steps, fused minibatch step, `--timesteps` env steps (the quadrotor leaves the crash plateau after ~35 M and is > 90 % successful from ~40 M).
One factor of the vehicle is changed at a time between the reference quadrotor (learns) and the SDF-derived hexacopter (does not):

  quad              the reference vehicle (control)
  hexa              hexacopter as shipped: 2.72 kg, I = (0.044, 0.044, 0.077), rotor thrust 0.47 .. 14.2 N, 1 N m per unit moment action
  hexa_sat          ... with the moment action scaled so that +-1 = the physical roll / pitch authority at hover (what the quadrotor's 0.1 N m is to it)
  hexa_nofloor      ... with rotor limits like the quadrotor's (0 .. 2 x hover thrust)
  hexa_sat_nofloor  both
  hexa_agile        ... and the inertia scaled down to the quadrotor's angular acceleration at saturation (a hexacopter as agile as the quadrotor)
  quad_floor        the quadrotor with the hexacopter's relative rotor limits (floor 10.6 % of hover, cap 3.2 x hover)
  hexa_bc           hexacopter as shipped, actor initialised by behaviour cloning of PidWaypointPolicy (the HIP PID + minimum-snap baseline), log_std = -1
  hexa_ls-1         hexacopter as shipped, random actor, initial log_std = -1 (std 0.37 instead of SB3's 1.0): exploration noise alone
  hexa_bc_ls0       the cloned actor with SB3's initial log_std = 0: the warm start alone
  quad_ls-1         the quadrotor with initial log_std = -1 (control)
  hexa_hoverbias    hexacopter as shipped, random actor with the thrust bias at hover (action_net.bias[0] = 1): does leaving the zero-thrust free fall suffice?
  hexa_arm_dagger   the headline vehicle, actor cloned with three DAgger rounds (the student flies, the PID labels its states), log_std = -1
  hexa_arm_bc       the HEADLINE vehicle (hexacopter + arm, tool-point task) with the cloned actor (PID in tool mode, joints at home), log_std = -1

  python tools/hexa_learn_sweep.py --runs quad hexa ... --timesteps 60000000 --out gpurun_out/hexa_sweep.json
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def make_cfg(amd, name, n):
    base = "quad" if name.startswith("quad") else "hexa"
    cfg = amd._lib.default_config(base, n)
    v = cfg.vehicle
    nr = v.n_rotors
    hover = v.mass * v.g / nr
    note = {}
    if name in ("hexa_sat", "hexa_sat_nofloor", "hexa_agile"):
        # roll authority at hover: every rotor can give up (hover - t_min) or gain (t_max - hover); moment arm sum |y_r|
        arm = sum(abs(v.mix[1 * nr + r]) for r in range(nr))
        lo = hover - (0.0 if name != "hexa_sat" else v.t_min[0])
        v.moment_scale = arm * lo
        note["moment_scale"] = v.moment_scale
    if name in ("hexa_nofloor", "hexa_sat_nofloor", "hexa_agile"):
        for r in range(nr):
            v.t_min[r] = 0.0; v.t_max[r] = 2.0 * hover
    if name == "hexa_agile":
        # quadrotor: saturation moment 2 * 0.086 * 0.44 N = 0.076 N m on 2.5e-4 kg m^2 -> ~300 rad/s^2; scale the inertia to reach that
        arm = sum(abs(v.mix[1 * nr + r]) for r in range(nr))
        target = 300.0
        s = (arm * hover / target) / v.inertia[0]
        for k in range(9):
            v.inertia[k] *= s; v.inv_inertia[k] /= s
        note["inertia_scale"] = s
    if name == "quad_floor":
        for r in range(nr):
            v.t_min[r] = 0.106 * hover; v.t_max[r] = 3.2 * hover
    cfg.seed = 0
    return cfg, note


def behaviour_clone(amd, torch, env, pol, steps=600, epochs=400, dagger_rounds=0):
    return amd.clone_pid_policy(env, pol, steps=steps, epochs=epochs, dagger_rounds=dagger_rounds)   # (the package's warm start: ppo.py)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", nargs="+", default=["quad", "hexa", "hexa_sat", "hexa_nofloor", "hexa_sat_nofloor", "hexa_agile", "quad_floor", "hexa_bc"])
    ap.add_argument("--timesteps", type=int, default=60_000_000)
    ap.add_argument("--envs", type=int, default=256)
    ap.add_argument("--n-steps", type=int, default=512)
    ap.add_argument("--out", default="gpurun_out/hexa_sweep.json")
    a = ap.parse_args()
    import torch
    import rl_aerial_manipulator_amd as amd
    res = {"what": __doc__.split("\n\n")[0], "timesteps": a.timesteps, "envs": a.envs, "n_steps": a.n_steps, "runs": {}}
    for name in a.runs:
        bc = name in ("hexa_bc", "hexa_bc_ls0", "hexa_arm_bc", "hexa_arm_dagger")
        if name.startswith("hexa_arm"):
            cfg, note = amd._lib.default_config("hexa_arm", a.envs), {}
            cfg.seed = 0
        else:
            cfg, note = make_cfg(amd, {"hexa_bc": "hexa", "hexa_bc_ls0": "hexa", "hexa_ls-1": "hexa", "quad_ls-1": "quad", "hexa_hoverbias": "hexa"}.get(name, name), a.envs)
        env = amd.GpuWaypointEnv(a.envs, config=cfg)
        model = amd.PPO(env, learning_rate=2e-4, n_steps=a.n_steps, batch_size=a.envs * a.n_steps // 128, n_epochs=12, gamma=0.995, gae_lambda=0.9, clip_range=0.2,
                        ent_coef=5e-4)
        if name == "hexa_hoverbias":
            with torch.no_grad():
                model.policy.action_net.bias.data[0] = 1.0
            note["action_bias_thrust"] = 1.0
        if name.endswith("ls-1"):
            with torch.no_grad():
                model.policy.log_std.data.fill_(-1.0)
            note["initial_log_std"] = -1.0
        if bc:
            note["bc_mse"] = behaviour_clone(amd, torch, env, model.policy, dagger_rounds=3 if name.endswith("dagger") else 0)
            if name == "hexa_bc_ls0":
                with torch.no_grad():
                    model.policy.log_std.data.fill_(0.0)
            note["initial_log_std"] = float(model.policy.log_std.data[0])
            env.reset()
            st = None
            env.stats(reset=True)
            obs = env.reset()
            for _ in range(2500):
                obs = env.step(model.policy.predict(obs))[0]
            st = env.stats(reset=True)
            note["bc_policy_success_rate_before_ppo"] = st["success"] / max(1, st["episodes"])
            model._started = False
        t0 = time.time()
        curve = []
        model.learn(a.timesteps, log_fn=lambda r: curve.append((r["timesteps"], r["ep_rew_mean"], r["ep_len_mean"], r["success_rate"])))
        dt = time.time() - t0
        tail = curve[-10:]
        first90 = next((c[0] for c in curve if c[3] > 0.9), None)
        first50 = next((c[0] for c in curve if c[3] > 0.5), None)
        v = cfg.vehicle
        res["runs"][name] = {"vehicle": {"mass": v.mass, "inertia_diag": [v.inertia[0], v.inertia[4], v.inertia[8]], "moment_scale": v.moment_scale, "n_rotors": v.n_rotors,
                                         "t_min": v.t_min[0], "t_max": v.t_max[0], "hover_thrust_per_rotor": v.mass * v.g / v.n_rotors, **note},
                             "seconds": dt, "iterations": len(curve),
                             "final_success_rate_mean_of_last_10_iterations": sum(c[3] for c in tail) / len(tail),
                             "final_ep_rew_mean": sum(c[1] for c in tail) / len(tail), "final_ep_len_mean": sum(c[2] for c in tail) / len(tail),
                             "timesteps_to_50pct_success": first50, "timesteps_to_90pct_success": first90,
                             "curve_every_10th_iteration": [dict(timesteps=c[0], ep_rew_mean=round(c[1], 1), ep_len_mean=round(c[2], 1), success_rate=round(c[3], 3)) for c in curve[::10]]}
        r = res["runs"][name]
        print(f"{name:18s} {dt:6.1f} s  success(last 10 it) {r['final_success_rate_mean_of_last_10_iterations']:.3f}  ep_len {r['final_ep_len_mean']:.0f}  ep_rew {r['final_ep_rew_mean']:.0f}"
              f"  50% at {first50}  90% at {first90}  {note}", flush=True)
        env.close()
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
