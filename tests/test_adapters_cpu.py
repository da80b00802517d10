"""Host logic of the SB3 VecEnv duck-type and of the single-env facade, CPU only: the backend is the test-only
OracleBackend, so what is checked here is the adapters' own code (info dicts, auto-reset bookkeeping, Monitor
episode records, the attribute surface the reference's visualiser reads) against the golden vectors."""
import os
import sys

import pytest
import numpy as np

import rl_aerial_manipulator_amd as amd
from tests import golden_util as G
from tests.oracle_backend import OracleBackend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rl-aerial-manipulator_amd", "compat"))


def info_from_bits(b):
    d = {}
    if b & 4: d.update(success=True, stopped=bool(b & 8))
    elif b & 16: d.update(success=False, crashed=True)
    elif b & 32: d.update(success=False, out_of_bounds=True)
    return d


def test_spaces_match_reference():
    obs, act = amd.vec_env.make_spaces()
    assert obs.shape == (20,) and obs.dtype == np.float32 and np.isinf(obs.low).all()
    assert np.array_equal(act.low, [0, -1, -1, -1]) and np.array_equal(act.high, [2, 1, 1, 1]) and act.dtype == np.float32


def test_vec_env_contract():
    n = 96
    ve = amd.GpuVecEnv(backend=OracleBackend(n, seed=4, max_episode_steps=150))
    assert ve.num_envs == n and ve.env_is_wrapped(object) == [False] * n and ve.get_attr("render_mode") == [None] * n
    obs = ve.reset()
    assert obs.shape == (n, 20) and obs.dtype == np.float32
    rng = np.random.RandomState(1)
    ret = np.zeros(n); length = np.zeros(n, int); n_done = 0; kinds = set()
    for t in range(400):
        a = rng.uniform([0, -1, -1, -1], [2, 1, 1, 1], (n, 4)).astype(np.float32)
        a[:, 1:] *= 0.05
        a[::6, 0] = 0.0  # some free-fall -> crashes; the 150-step limit -> TimeLimit truncations
        ve.step_async(a)
        obs, rew, done, infos = ve.step_wait()
        assert obs.shape == (n, 20) and rew.dtype == np.float32 and done.dtype == bool and len(infos) == n
        ret += rew; length += 1
        for i in range(n):
            if done[i]:
                n_done += 1
                inf = infos[i]
                assert set(inf) >= {"terminal_observation", "TimeLimit.truncated", "episode"}
                assert inf["episode"]["l"] == length[i] and abs(inf["episode"]["r"] - ret[i]) < 1e-2 * max(1, abs(ret[i]))
                assert inf["terminal_observation"].shape == (20,) and not np.array_equal(inf["terminal_observation"], obs[i])
                assert (obs[i, 3:6] == 0).all() and obs[i, 6] == 1  # the returned obs is the RESET observation
                kinds.add("crashed" if inf.get("crashed") else ("trunc" if inf["TimeLimit.truncated"] else "other"))
                ret[i] = 0; length[i] = 0
            else:
                assert "terminal_observation" not in infos[i] and "episode" not in infos[i]
    assert n_done > 50 and {"crashed", "trunc"} <= kinds
    ve.close()


def test_facade_replays_golden_episode_like_the_reference():
    from rl_env_scaledObs import WaypointQuadEnv
    d = G.load("policy_ep3")
    T = d["actions"].shape[0]
    be = OracleBackend(1, auto_reset=False)
    env = WaypointQuadEnv(backend=be)
    obs, info = env.reset()
    assert obs.shape == (20,) and obs.dtype == np.float32 and info == {}
    # put the facade's env on the golden episode's start
    f, i = be.get_state()
    f, i = f.numpy().copy(), i.numpy().copy()
    G.fill_blob(f, i, {**d, "actions": d["actions"][:1]})
    be.set_state(f, i)
    env._cache = None
    assert np.allclose(env.quadcopter.state, d["state"][0]) and np.allclose(env.waypoint_list[0], d["waypoints"][0])
    assert np.array_equal(env._get_observation(), d["obs0"])
    for t in range(T):
        obs, r, term, trunc, info = env.step(d["actions"][t])
        assert isinstance(r, float) and isinstance(term, bool) and isinstance(trunc, bool)
        assert env.F == d["F"][t] and np.array_equal(env.M, d["M"][t])          # telemetry: float32 products
        assert (term, trunc) == (bool(d["terminated"][t]), bool(d["truncated"][t]))
        assert info == info_from_bits(int(d["info_bits"][t])), (t, info)
        assert np.abs(obs - d["obs"][t]).max() < 2e-3 and abs(r - d["reward"][t]) < 2e-2 * max(1, abs(d["reward"][t]))  # free-running
        assert env.waypoint_index == d["var_waypoint_index"][t + 1] and env.counter == d["var_counter"][t + 1]
        assert env.current_step == t + 1 and env.final_waypoint_reached == bool(d["var_fwr"][t + 1])
    assert term and env.counter == 501
    wf = env.quadcopter.world_frame()
    assert wf.shape == (3, 6) and np.allclose(wf[:, 4], env.quadcopter.position())
    assert np.allclose(np.linalg.norm(wf[:, 0] - wf[:, 4]), 0.086) and len(env.quadcopter.attitude()) == 3
    assert np.allclose(env.current_waypoint, d["waypoints"][0]) and abs(env.final_yaw - float(d["final_yaw"])) < 1e-12


@pytest.mark.parametrize("module,variant,gold", [("rl_env", "TASK_V1_RAW17", "v1r_reach2"), ("rl_env_scaledObs", "TASK_V1_SCALED17", "v1s_reach2")])
def test_v1_facades_replay_golden_episodes(module, variant, gold):
    """`from rl_env import WaypointQuadEnv` (v1/rl_train_vecN.py:5) / `from rl_env_scaledObs import ...` (v1/rl_train.py:4)
    resolve to compat/v1/: 17-D observations, two waypoints in this episode, +400 termination on the last one."""
    import importlib.util
    from oracle import oracle as O
    path = os.path.join(ROOT, "rl-aerial-manipulator_amd", "compat", "v1", module + ".py")
    spec = importlib.util.spec_from_file_location("_v1_facade_" + module, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    d = G.load(gold)
    be = OracleBackend(1, auto_reset=False, variant=getattr(O, variant))
    env = mod.WaypointQuadEnv(backend=be)
    assert env.TASK.startswith("v1") and env.observation_space.shape == (17,)
    obs, info = env.reset()
    assert obs.shape == (17,) and obs.dtype == np.float32 and info == {} and len(env.waypoint_list) in (1, 2)
    f, i = be.get_state()
    f, i = f.numpy().copy(), i.numpy().copy()
    G.fill_blob(f, i, {**d, "actions": d["actions"][:1]}, per_env_k=True)
    be.set_state(f, i)
    env._cache = None
    assert len(env.waypoint_list) == 2 and np.allclose(env.waypoint_list[1], d["waypoints"][1])
    assert np.array_equal(env._get_observation(), d["obs0"])
    T = d["actions"].shape[0]
    for t in range(T):
        obs, r, term, trunc, info = env.step(d["actions"][t])
        assert (term, trunc) == (bool(d["terminated"][t]), bool(d["truncated"][t]))
        assert np.abs(obs - d["obs"][t]).max() < 5e-3 * max(1.0, np.abs(d["obs"][t]).max())      # free-running
        assert env.waypoint_index == d["var_waypoint_index"][t + 1] and env.current_step == d["var_current_step"][t + 1]
    assert term and info.get("success") is True
