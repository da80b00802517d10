#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the UNMODIFIED reference.

Runs only in the build container (needs /root/reference).  Nothing from the reference
is copied: the reference modules are imported from where they lie, driven with inputs
chosen here, and only inputs + observed outputs are written out as .npz data.

The reference env (`initial-implementation-v2/rl_env_scaledObs.py`) needs `gymnasium`
only for a base class and `spaces.Box`; neither is installed here, so a minimal
in-memory stand-in for those two names is registered before the import (our own code;
see SURVEY.md App. F).  Everything numerical -- numpy, scipy.integrate.odeint,
scipy Rotation -- is the real thing.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [--out tests/golden]
"""
import argparse
import contextlib
import io
import json
import os
import sys
import types
import zipfile

import numpy as np

REF_V2 = "/root/reference/initial-implementation-v2"
CKPT = os.path.join(REF_V2, "checkpoints_from_8_6M", "ppo_model_2300000_steps.zip")

# info_bits layout shared with include/amenv.h
BIT_TERMINATED, BIT_TRUNCATED, BIT_SUCCESS, BIT_STOPPED, BIT_CRASHED, BIT_OOB = (1 << i for i in range(6))


def _install_gymnasium_stub():
    gym = types.ModuleType("gymnasium")
    sp = types.ModuleType("gymnasium.spaces")

    class Env:  # only what the reference env touches
        def reset(self, seed=None, options=None):
            return None

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
            self.shape = shape or self.low.shape
            self.dtype = dtype

    gym.Env, sp.Box, gym.spaces = Env, Box, sp
    sys.modules["gymnasium"], sys.modules["gymnasium.spaces"] = gym, sp


def _import_reference(root=REF_V2, module="rl_env_scaledObs"):
    sys.dont_write_bytecode = True
    _install_gymnasium_stub()
    sys.path.insert(0, root)
    import importlib
    with contextlib.redirect_stdout(io.StringIO()):
        WaypointQuadEnv = importlib.import_module(module).WaypointQuadEnv  # unmodified reference file
        from simul_files.model.quadcopter import Quadcopter  # noqa: E402
    return WaypointQuadEnv, Quadcopter


class CheckpointPolicy:
    """Deterministic SB3 MlpPolicy forward, rebuilt from policy.pth tensors (weights_only load)."""

    def __init__(self, path):
        import torch

        self.obs_mean = None  # set for VecNormalize-trained policies
        with zipfile.ZipFile(path) as z:
            sd = torch.load(io.BytesIO(z.read("policy.pth")), weights_only=True, map_location="cpu")
        g = lambda k: sd[k].double().numpy()
        self.layers = [(g(f"mlp_extractor.policy_net.{i}.weight"), g(f"mlp_extractor.policy_net.{i}.bias")) for i in (0, 2, 4)]
        self.head = (g("action_net.weight"), g("action_net.bias"))
        self.low = np.array([0, -1, -1, -1], dtype=np.float32)
        self.high = np.array([2, 1, 1, 1], dtype=np.float32)

    def __call__(self, obs):
        h = obs.astype(np.float64)
        for w, b in self.layers:
            h = np.tanh(w @ h + b)
        a = (self.head[0] @ h + self.head[1]).astype(np.float32)
        return np.clip(a, self.low, self.high)


def env_vars(env):
    """Scalar per-episode variables of the reference env; the v1 envs lack the hold-phase fields."""
    return dict(
        waypoint_index=int(env.waypoint_index),
        last_distance=float("nan") if env.last_distance is None else float(env.last_distance),
        fwr=bool(getattr(env, "final_waypoint_reached", False)),
        counter=int(getattr(env, "counter", 0) or 0),
        counter_activated=bool(getattr(env, "counter_activated", False)),
        current_step=int(env.current_step),
    )


def info_to_bits(terminated, truncated, info):
    b = 0
    b |= BIT_TERMINATED if terminated else 0
    b |= BIT_TRUNCATED if truncated else 0
    b |= BIT_SUCCESS if info.get("success", False) else 0
    b |= BIT_STOPPED if info.get("stopped", False) else 0
    b |= BIT_CRASHED if info.get("crashed", False) else 0
    b |= BIT_OOB if info.get("out_of_bounds", False) else 0
    return b


def run_episode(env, action_fn, max_steps, stop_on_done=True):
    """Step the reference env; record everything needed to teacher-force each step.

    Index convention: state[t], vars[t] are the values BEFORE step t (t=0: after reset);
    state[t+1] are the values after step t.  obs/reward/flags[t] are step t's outputs.
    """
    rec = dict(state=[env.quadcopter.state.copy()], actions=[], obs=[], reward=[], terminated=[], truncated=[], info_bits=[], F=[], M=[])
    vars_ = [env_vars(env)]
    obs = env._get_observation()
    obs0 = obs.copy()
    sink = io.StringIO()
    for t in range(max_steps):
        a = np.asarray(action_fn(t, obs, env), dtype=np.float32)
        with contextlib.redirect_stdout(sink):
            obs, r, term, trunc, info = env.step(a)
        rec["actions"].append(a)
        rec["obs"].append(obs.copy())
        rec["reward"].append(float(r))
        rec["terminated"].append(bool(term))
        rec["truncated"].append(bool(trunc))
        rec["info_bits"].append(info_to_bits(term, trunc, info))
        rec["F"].append(float(getattr(env, "F", np.nan) if getattr(env, "F", None) is not None else np.nan))
        rec["M"].append(np.asarray(getattr(env, "M", None) if getattr(env, "M", None) is not None else [np.nan] * 3, dtype=np.float64).copy())
        rec["state"].append(env.quadcopter.state.copy())
        vars_.append(env_vars(env))
        if stop_on_done and (term or trunc):
            break
    out = dict(
        state=np.asarray(rec["state"], dtype=np.float64),
        actions=np.asarray(rec["actions"], dtype=np.float32),
        obs=np.asarray(rec["obs"], dtype=np.float32),
        obs0=obs0.astype(np.float32),
        reward=np.asarray(rec["reward"], dtype=np.float64),
        terminated=np.asarray(rec["terminated"], dtype=np.bool_),
        truncated=np.asarray(rec["truncated"], dtype=np.bool_),
        info_bits=np.asarray(rec["info_bits"], dtype=np.uint32),
        F=np.asarray(rec["F"], dtype=np.float64),
        M=np.asarray(rec["M"], dtype=np.float64),
        waypoints=np.asarray(env.waypoint_list, dtype=np.float64),
        final_yaw=np.float64(getattr(env, "final_yaw", 0.0) or 0.0),
    )
    for k in vars_[0]:
        out["var_" + k] = np.asarray([v[k] for v in vars_])
    return out


def fresh_env(WaypointQuadEnv, seed, start=None, waypoint=None, final_yaw=None):
    np.random.seed(seed)
    env = WaypointQuadEnv()
    with contextlib.redirect_stdout(io.StringIO()):
        env.reset()
    if start is not None:
        env.quadcopter.state[0:3] = np.asarray(start, dtype=np.float64)
    if waypoint is not None:
        wps = np.asarray(waypoint, dtype=np.float64).reshape(-1, 3)
        env.waypoint_list = [w.copy() for w in wps]
        env.num_waypoints = len(env.waypoint_list)
        env.current_waypoint = env.waypoint_list[0]
    if final_yaw is not None:
        env.final_yaw = float(final_yaw)
    return env


def hover_pd(target):
    """A small PD position/attitude hold (our own code) -> bounded, information-rich actions."""
    target = np.asarray(target, dtype=np.float64)

    def fn(t, obs, env):
        s = env.quadcopter.state
        pos, vel, q, w = s[0:3], s[3:6], s[6:10], s[10:13]
        acc = 6.0 * (target - pos) - 4.0 * vel
        thrust = 1.0 + acc[2] / 9.81
        # small-angle desired roll/pitch from lateral accel
        # in this model +roll (qx>0) accelerates +y and +pitch (qy>0) accelerates -x (third row of R(q))
        roll_d, pitch_d = acc[1] / 9.81, -acc[0] / 9.81
        roll = np.arctan2(2 * (q[0] * q[1] + q[2] * q[3]), 1 - 2 * (q[1] ** 2 + q[2] ** 2))
        pitch = np.arcsin(np.clip(2 * (q[0] * q[2] - q[3] * q[1]), -1, 1))
        yaw = np.arctan2(2 * (q[0] * q[3] + q[1] * q[2]), 1 - 2 * (q[2] ** 2 + q[3] ** 2))
        # qdot = -1/2 Omega(w) q  =>  d(roll)/dt = -p: attitude error feeds back with a minus sign
        mx = -0.5 * (np.clip(roll_d, -0.4, 0.4) - roll) - 0.06 * w[0]
        my = -0.5 * (np.clip(pitch_d, -0.4, 0.4) - pitch) - 0.06 * w[1]
        mz = -0.3 * (0.0 - yaw) - 0.08 * w[2]
        a = np.array([thrust, mx, my, mz]) + 0.01 * np.sin(0.05 * t + np.arange(4))
        return np.clip(a, [0, -1, -1, -1], [2, 1, 1, 1])

    return fn


REF_V1 = "/root/reference/initial-implementation-v1"


def main_v1(out, module, tag):
    """Golden vectors of the v1 envs (17-D obs; `rl_env_scaledObs.py` scaled, `rl_env.py` raw).  Separate process per
    module: v1 and v2 share module names."""
    WaypointQuadEnv, _ = _import_reference(REF_V1, module)
    files = {}

    def save(name, d):
        np.savez_compressed(os.path.join(out, name + ".npz"), **d)
        files[name] = dict(steps=int(d["actions"].shape[0]))
        print(f"  {name}: T={d['actions'].shape[0]} last bits={int(d['info_bits'][-1])} K={d['waypoints'].shape[0]} return={d['reward'].sum():.1f}")

    # PD flight through one / two waypoints: approach shaping (+-10), intermediate waypoint, final reach (+400, bonuses)
    def chase(t, obs, env):
        return hover_pd(env.current_waypoint)(t, obs, env)

    env = fresh_env(WaypointQuadEnv, 1, start=[0.0, 0.0, 1.5], waypoint=[[0.5, 0.3, 1.8]])
    save(f"{tag}_reach1", run_episode(env, chase, 1300))
    env = fresh_env(WaypointQuadEnv, 2, start=[0.2, -0.4, 1.2], waypoint=[[0.6, 0.1, 1.6], [-0.3, 0.5, 2.2]])
    save(f"{tag}_reach2", run_episode(env, chase, 1300))
    # fast fly-through: reaches the final waypoint moving (negative stop bonuses)
    env = fresh_env(WaypointQuadEnv, 3, start=[0.0, 0.0, 1.5], waypoint=[[0.0, 0.0, 2.4]])
    save(f"{tag}_flythrough", run_episode(env, lambda t, o, e: np.array([1.6, 0.0, 0.0, 0.01]), 400))
    env = fresh_env(WaypointQuadEnv, 4, start=[0.1, 0.1, 1.2], waypoint=[[0.5, 0.5, 2.0]])
    save(f"{tag}_crash", run_episode(env, lambda t, o, e: np.array([0.0, 0.01, -0.01, 0.0]), 400))
    env = fresh_env(WaypointQuadEnv, 5, start=[0.0, 0.0, 1.5], waypoint=[[-0.8, 0.6, 1.2], [0.3, 0.3, 2.5]])
    save(f"{tag}_oob", run_episode(env, lambda t, o, e: np.array([2.0, 0.02 if t < 10 else 0.0, 0.0, 0.0]), 1300))
    env = fresh_env(WaypointQuadEnv, 6, start=[0.0, 0.0, 1.5], waypoint=[[0.9, -0.9, 2.8]])
    d = run_episode(env, hover_pd([0.0, 0.0, 1.5]), 1300)
    save(f"{tag}_timelimit", d)
    print(f"    truncated first at {int(np.argmax(d['truncated']))}")
    if tag == "v1s":
        n = 10000
        np.random.seed(4321)
        env = WaypointQuadEnv()
        start = np.zeros((n, 3), np.float32); wp = np.full((n, 2, 3), np.nan, np.float32); K = np.zeros(n, np.uint8)
        for i in range(n):
            env.reset()
            start[i] = env.quadcopter.state[0:3]; K[i] = len(env.waypoint_list)
            wp[i, :K[i]] = np.asarray(env.waypoint_list)
        np.savez_compressed(os.path.join(out, "v1_reset_samples.npz"), start=start, waypoints=wp, K=K)
        files["v1_reset_samples"] = dict(steps=n)
    with open(os.path.join(out, f"META_{tag}.json"), "w") as f:
        json.dump(dict(generator=f"tools/gen_golden.py --v1 {module}", reference=f"initial-implementation-v1/{module}.py (unmodified, imported in place)",
                       numpy=np.__version__, files=files), f, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--v1", default=None, choices=["rl_env_scaledObs", "rl_env"], help="record the v1 env of that module instead of v2")
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    if args.v1:
        return main_v1(out, args.v1, "v1s" if args.v1 == "rl_env_scaledObs" else "v1r")

    WaypointQuadEnv, Quadcopter = _import_reference()
    import scipy

    meta = dict(
        generator="tools/gen_golden.py",
        reference="LahiruCooray/rl-aerial-manipulator initial-implementation-v2 (unmodified, imported in place)",
        numpy=np.__version__, scipy=scipy.__version__, python=sys.version.split()[0],
        note="state[t]/var_*[t] = before step t; state[t+1] = after step t; obs/reward/flags[t] = outputs of step t",
        info_bits="bit0 terminated, 1 truncated, 2 success, 3 stopped, 4 crashed, 5 out_of_bounds",
    )
    files = {}

    def save(name, d):
        np.savez_compressed(os.path.join(out, name + ".npz"), **d)
        files[name] = dict(steps=int(d["actions"].shape[0]) if "actions" in d else None)
        print(f"  {name}: " + ", ".join(f"{k}{tuple(np.shape(v))}" for k, v in d.items() if np.ndim(v) > 0)[:200])

    # (i) policy-driven full episodes: reach -> hold -> terminate
    policy = CheckpointPolicy(CKPT)
    for seed in range(4):
        env = fresh_env(WaypointQuadEnv, seed)
        d = run_episode(env, lambda t, obs, env: policy(obs), 2100)
        save(f"policy_ep{seed}", d)
        print(f"    seed {seed}: T={len(d['reward'])} return={d['reward'].sum():.1f} last bits={int(d['info_bits'][-1])}")

    # (ii) 1000-step gentle open-loop sequence that stays in bounds without reaching (config-1 trace)
    env = fresh_env(WaypointQuadEnv, 100, start=[0.2, -0.3, 1.5], waypoint=[0.9, 0.8, 2.6], final_yaw=0.7)
    rng = np.random.RandomState(7)
    ph = rng.uniform(0, 2 * np.pi, 4)

    def gentle(t, obs, env):
        # zero-mean wiggles: each moment term integrates to ~0 tilt so the open-loop quad stays in bounds
        return np.array([1.0 + 0.05 * np.cos(0.05 * t + ph[0]), 0.002 * np.cos(0.25 * t) * np.sign(np.sin(0.02 * t + ph[1])),
                         0.002 * np.cos(0.2 * t) * np.sign(np.sin(0.017 * t + ph[2])), 0.02 * np.cos(0.05 * t + ph[3])])

    save("openloop_1000", run_episode(env, gentle, 1000))

    # (iii) crash: zero thrust -> falls through z<0.1; falling-penalty branch
    env = fresh_env(WaypointQuadEnv, 101, start=[0.1, 0.1, 1.2], waypoint=[0.5, 0.5, 2.0])
    save("crash", run_episode(env, lambda t, o, e: np.array([0.0, 0.01, -0.01, 0.0]), 400))

    # (iv) out of bounds: full thrust with a tilt
    env = fresh_env(WaypointQuadEnv, 102, start=[0.0, 0.0, 1.5], waypoint=[-0.8, 0.6, 0.8])
    save("oob", run_episode(env, lambda t, o, e: np.array([2.0, 0.02 if t < 10 else 0.0, 0.0, 0.0]), 2100))

    # (v) time limit: PD hold away from the waypoint for > 2000 steps -> truncated on call 2001
    env = fresh_env(WaypointQuadEnv, 103, start=[0.0, 0.0, 1.5], waypoint=[0.9, -0.9, 2.8])
    d = run_episode(env, hover_pd([0.0, 0.0, 1.5]), 2100)
    save("timelimit", d)
    print(f"    timelimit: T={len(d['reward'])} truncated first at {int(np.argmax(d['truncated']))}")

    # (vi) motor saturation: large moments/thrust so the per-prop clamp binds
    env = fresh_env(WaypointQuadEnv, 104, start=[0.0, 0.0, 1.8], waypoint=[0.7, 0.7, 2.5])
    rng = np.random.RandomState(11)
    sat_actions = rng.uniform([0, -1, -1, -1], [2, 1, 1, 1], size=(300, 4))
    save("saturation", run_episode(env, lambda t, o, e: sat_actions[t], 300))

    # (vi-b) PD flight that reaches the waypoint, leaves and returns
    env = fresh_env(WaypointQuadEnv, 105, start=[0.0, 0.0, 1.5], waypoint=[0.4, 0.0, 1.6], final_yaw=-1.0)
    # hold 400 steps, leave (counter keeps running outside the ball), come back with counter > limit -> terminate on re-entry
    tgt = lambda t: [0.4, 0.6, 1.6] if 500 <= t < 900 else [0.4, 0.0, 1.6]
    save("reach_and_leave", run_episode(env, lambda t, o, e: hover_pd(tgt(t))(t, o, e), 1300))

    # (vii) single-step dynamics pairs: random (state, F, M) -> Quadcopter.update -> next state
    rng = np.random.RandomState(3)
    n = 2048
    s_in = np.zeros((n, 13)); s_out = np.zeros((n, 13)); a_in = np.zeros((n, 4), dtype=np.float32)
    quad = Quadcopter([0, 0, 1], (0, 0, 0))
    for i in range(n):
        q = rng.normal(size=4); q /= np.linalg.norm(q); q *= np.sign(q[0]) if q[0] != 0 else 1.0
        s = np.concatenate([rng.uniform(-2, 2, 3), rng.normal(0, 1.5, 3), q, rng.normal(0, 3.0, 3)])
        a = rng.uniform([0, -1, -1, -1], [2, 1, 1, 1]).astype(np.float32)
        F = a[0] * np.float64(0.18) * np.float64(9.81) if False else a[0] * 0.18 * 9.81  # same expression as the env (float32 under NumPy>=2)
        M = a[1:4] * 0.1
        quad.state = s.copy()
        quad.update(1.0 / 200.0, F, M.reshape(-1, 1))
        s_in[i], s_out[i], a_in[i] = s, quad.state, a
    np.savez_compressed(os.path.join(out, "dynamics_pairs.npz"), state_in=s_in, actions=a_in, state_out=s_out)
    files["dynamics_pairs"] = dict(steps=n)

    # (viii) reset distribution: 10k resets of the reference env
    n = 10000
    np.random.seed(12345)
    env = WaypointQuadEnv()
    start = np.zeros((n, 3), np.float32); wp = np.zeros((n, 3), np.float32); fy = np.zeros(n, np.float32); kind = np.zeros(n, np.uint8)
    names = {"Linear": 0, "Curved-Z": 1, "Curved-Y": 2, "Curved-X": 3, "Helical": 4}
    for i in range(n):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            env.reset()
        start[i] = env.quadcopter.state[0:3]; wp[i] = env.waypoint_list[0]; fy[i] = env.final_yaw
        kind[i] = names[buf.getvalue().strip().splitlines()[-1]]
        assert np.array_equal(env.quadcopter.state[3:], [0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
    np.savez_compressed(os.path.join(out, "reset_samples.npz"), start=start, waypoint=wp, final_yaw=fy, kind=kind)
    files["reset_samples"] = dict(steps=n, kinds=names)

    meta["files"] = files
    with open(os.path.join(out, "META.json"), "w") as f:
        json.dump(meta, f, indent=1)
    tot = sum(os.path.getsize(os.path.join(out, x)) for x in os.listdir(out))
    print(f"wrote {len(files)} fixtures, {tot/1e6:.2f} MB, to {out}")


if __name__ == "__main__":
    main()
