"""Pin the CPU oracle (oracle/amenv_oracle.c) to the golden vectors made by the unmodified reference.

CPU-only.  Tolerances: everything but the ODE solve is an exact restatement (fp64 rounding);
RK4(dt = 5 ms) vs the reference's LSODA differs by < 1e-6 abs per teacher-forced step
(measured ~5e-8 = the size of LSODA's own error at its default rtol/atol = 1.49e-8).
"""
import numpy as np
import pytest

from oracle import oracle as O
from tests import golden_util as G

STATE_ATOL = 1e-6      # RK4 vs LSODA, teacher-forced per step (4.8e-8 measured)
OBS_ATOL = 2e-7        # obs are the state scaled by <= 1 and cast to f32
OBS_RTOL = 2e-7
REWARD_ATOL = 5e-5     # reward has 20*(d_prev - d) and 100*vz terms amplifying the state difference


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        got = O.philox(key[0] | (key[1] << 32), ctr[0] | (ctr[1] << 32), ctr[2], ctr[3])
        assert tuple(int(x) for x in got) == want


def test_reference_constants():
    cfg = O.reference_quad_config()
    v = cfg.vehicle
    assert v.n_rotors == 4 and v.mass == 0.18 and v.g == 9.81
    A = np.array(v.mix[:16]).reshape(4, 4)
    invA = np.array(v.alloc[:16]).reshape(4, 4)
    np.testing.assert_allclose(A @ invA, np.eye(4), atol=1e-13)
    np.testing.assert_allclose(np.array(v.inertia).reshape(3, 3) @ np.array(v.inv_inertia).reshape(3, 3), np.eye(3), atol=1e-12)
    assert abs(v.t_max[0] - 0.88290) < 1e-12 and v.t_min[0] == 0.0
    # 1/(2L) = 5.8139535, 1/(4r) = 10.1833333
    assert abs(invA[1, 1] - 5.813953488372093) < 1e-9 and abs(invA[0, 3] - 10.183333333333334) < 1e-9


def test_dynamics_pairs_vs_reference():
    """Quadcopter.update on 2048 random (state, action) pairs: mixer+clamp exact, RK4 vs LSODA < 1e-6."""
    d = G.load("dynamics_pairs")
    cfg = O.reference_quad_config()
    err = 0.0
    for s_in, a, s_out in zip(d["state_in"], d["actions"], d["state_out"]):
        s, _ = O.dynamics_step(cfg, s_in, a)
        err = max(err, np.abs(s - s_out).max())
    assert err < STATE_ATOL, err


@pytest.mark.parametrize("name", G.EPISODES)
def test_teacher_forced_episode(name):
    """Every step of every golden episode, started from the reference's own pre-step state."""
    d = G.load(name)
    T = d["actions"].shape[0]
    cfg = O.reference_quad_config(num_envs=T, flags=0)  # raw env semantics: no auto-reset
    env = O.OracleEnv(cfg)
    G.fill_blob(env.fstate, env.istate, d)
    out = env.step(d["actions"])
    # dynamics
    s_err = np.abs(env.fstate[0:13].T - d["state"][1:]).max()
    assert s_err < STATE_ATOL, s_err
    # telemetry F, M of the golden episodes = the f32 action scaling (rl_env_scaledObs.py:125-128)
    for t in (0, T // 2, T - 1):
        _, w = O.dynamics_step(cfg, d["state"][t], d["actions"][t])
        assert w[0] == d["F"][t] and np.array_equal(w[1:4], d["M"][t])
    # flags / info / state machine: exact
    assert np.array_equal(out["info"] & 63, d["info_bits"]), np.nonzero((out["info"] & 63) != d["info_bits"])
    assert np.array_equal(out["done"].astype(bool), d["terminated"] | d["truncated"])
    assert np.array_equal(env.istate[O.I_STEP], d["var_current_step"][1:])
    assert np.array_equal(env.istate[O.I_COUNTER], d["var_counter"][1:])
    fl = env.istate[O.I_FLAGS]
    assert np.array_equal(fl & 255, d["var_waypoint_index"][1:])
    assert np.array_equal((fl & O.FLAGBIT_FWR) != 0, d["var_fwr"][1:])
    assert np.array_equal((fl & O.FLAGBIT_COUNTER_ACTIVE) != 0, d["var_counter_activated"][1:])
    np.testing.assert_allclose(env.fstate[O.F_LAST_DISTANCE], d["var_last_distance"][1:], atol=STATE_ATOL)
    # observation and reward
    np.testing.assert_allclose(out["obs"], d["obs"], rtol=OBS_RTOL, atol=OBS_ATOL)  # a state diff of 5e-8 can flip one f32 ulp
    assert np.abs(out["reward"] - d["reward"]).max() < REWARD_ATOL, np.abs(out["reward"] - d["reward"]).max()
    # Monitor outputs on done steps
    for t in np.nonzero(out["done"])[0]:
        assert out["ep_len"][t] == d["var_current_step"][t + 1]
        assert np.array_equal(out["terminal_obs"][t], out["obs"][t])


@pytest.mark.parametrize("name", ["policy_ep0", "openloop_1000", "crash", "timelimit"])
def test_free_running_episode(name):
    """Open-loop replay from the reset state: event steps identical, drift reported (the quad is
    open-loop unstable and LSODA itself is 4.6e-5 off a tight-tolerance solve after ~800 steps)."""
    d = G.load(name)
    T = d["actions"].shape[0]
    cfg = O.reference_quad_config(num_envs=1, flags=0)
    env = O.OracleEnv(cfg)
    env.set_env(0, d["state"][0], d["waypoints"], float(d["final_yaw"]), None, 0, False, 0, False, 0)
    assert np.array_equal(env.observe()[0], d["obs0"])
    drift = 0.0
    bits = np.zeros(T, np.uint32)
    ret = 0.0
    for t in range(T):
        out = env.step(d["actions"][t:t + 1])
        bits[t] = out["info"][0] & 63
        ret += out["reward"][0]
        drift = max(drift, np.abs(env.fstate[0:13, 0] - d["state"][t + 1]).max())
    assert drift < 2e-3, drift
    # the state machine fires at the same steps as in the reference
    assert np.array_equal(bits & 3, d["info_bits"] & 3)
    assert np.count_nonzero(bits != d["info_bits"]) <= 2  # 'stopped' may flip on a threshold by drift
    assert abs(ret - d["reward"].sum()) < 1e-3 * max(1.0, abs(d["reward"].sum()))


def test_reset_distribution_vs_reference():
    """Philox reset spec vs 10k reset() calls of the reference (global MT19937): same distributions."""
    d = G.load("reset_samples")
    n = 20000
    cfg = O.reference_quad_config(num_envs=n, seed=2024)
    env = O.OracleEnv(cfg)
    obs = env.reset()
    st, wp, fy = env.fstate[0:3].T, env.fstate[O.F_WP0:O.F_WP0 + 3].T, env.fstate[O.F_FINAL_YAW]
    # exact support
    assert (np.abs(st[:, :2]) <= 1).all() and (st[:, 2] >= 1).all() and (st[:, 2] < 2).all()
    assert (np.abs(fy) <= np.pi + 1e-6).all()
    assert np.array_equal(env.fstate[3:13, 0], [0, 0, 0, 1, 0, 0, 0, 0, 0, 0])
    assert (env.istate[O.I_EPISODE] == 1).all() and (env.fstate[O.F_LAST_DISTANCE] == -1).all()
    # moments of start / yaw
    for a, b in ((st, d["start"]), (fy[:, None], d["final_yaw"][:, None])):
        assert np.abs(a.mean(0) - b.mean(0)).max() < 0.03
        assert np.abs(a.std(0) - b.std(0)).max() < 0.03
    # trajectory mix 0.30 / 0.42 / 0.28 (helical waypoint = start + [0.8, 0, 0.4] identifies itself)
    hel = np.abs((wp - st) - np.array([0.8, 0.0, 0.4])).max(1) < 1e-5
    ref_hel = d["kind"] == 4
    assert abs(hel.mean() - 0.28) < 0.015 and abs(ref_hel.mean() - 0.28) < 0.02
    # non-helical waypoints: x,y ~ U(-1,1), z ~ U(0.5,3) in both
    a, b = wp[~hel], d["waypoint"][~ref_hel]
    assert np.abs(a.mean(0) - b.mean(0)).max() < 0.04 and np.abs(a.std(0) - b.std(0)).max() < 0.03
    from scipy import stats
    for c in range(3):
        assert stats.ks_2samp(a[:, c], b[:, c]).pvalue > 1e-3
        assert stats.ks_2samp(st[:, c], d["start"][:, c]).pvalue > 1e-3
    # the reset observation is _get_observation of that state
    np.testing.assert_allclose(obs[:, 0:3], st / 10, atol=1e-7)
    np.testing.assert_allclose(obs[:, 13:16], (wp - st) / 2, atol=1e-7)
    assert (obs[:, 16:19] == 0).all()


def test_reset_is_keyed_by_global_env_id():
    """Sharding invariance: env g of an N-env job == env 0 of a shard with env_id_offset = g."""
    full = O.OracleEnv(O.reference_quad_config(num_envs=64, seed=7))
    full.reset()
    cfg = O.reference_quad_config(num_envs=16, seed=7)
    cfg.env_id_offset = 32
    shard = O.OracleEnv(cfg)
    shard.reset()
    assert np.array_equal(shard.fstate, full.fstate[:, 32:48])


def test_auto_reset_semantics():
    """DummyVecEnv contract: on done the returned obs is the reset obs, terminal_obs holds the last one."""
    d = G.load("crash")
    T = d["actions"].shape[0]
    cfg = O.reference_quad_config(num_envs=1, seed=3, flags=O.FLAG_AUTO_RESET)
    env = O.OracleEnv(cfg)
    env.set_env(0, d["state"][0], d["waypoints"], float(d["final_yaw"]), None, 0, False, 0, False, 0, episode=5)
    tot = 0.0
    for t in range(T):
        out = env.step(d["actions"][t:t + 1])
        tot += out["reward"][0]
    assert out["done"][0] == 1 and out["info"][0] & O.INFO_CRASHED and out["info"][0] & O.INFO_WAS_RESET
    assert out["ep_len"][0] == T and abs(out["ep_return"][0] - tot) < 1e-2
    assert np.abs(out["terminal_obs"][0] - d["obs"][-1]).max() < 1e-4
    assert env.istate[O.I_EPISODE, 0] == 6 and env.istate[O.I_STEP, 0] == 0
    assert np.array_equal(out["obs"][0], env.observe()[0])
    assert env.fstate[2, 0] >= 1.0  # fresh start height


# ---- v1 envs (17-D obs): v1/rl_env_scaledObs.py (scaled) and v1/rl_env.py (raw) --------------------------------------
@pytest.mark.parametrize("tag,variant", [("v1s", O.TASK_V1_SCALED17), ("v1r", O.TASK_V1_RAW17)])
@pytest.mark.parametrize("name", G.V1_EPISODES)
def test_v1_teacher_forced_episode(tag, variant, name):
    d = G.load(f"{tag}_{name}")
    T = d["actions"].shape[0]
    cfg = O.reference_quad_config(num_envs=T, flags=0, variant=variant)
    env = O.OracleEnv(cfg)
    assert env.obs_dim == 17
    G.fill_blob(env.fstate, env.istate, d, per_env_k=True)
    out = env.step(d["actions"])
    assert np.abs(env.fstate[0:13].T - d["state"][1:]).max() < STATE_ATOL
    assert np.array_equal(out["info"] & 63, d["info_bits"]), np.nonzero((out["info"] & 63) != d["info_bits"])
    assert np.array_equal(env.istate[O.I_STEP], d["var_current_step"][1:])       # final reach returns before the increment
    assert np.array_equal(env.istate[O.I_FLAGS] & 15, d["var_waypoint_index"][1:])
    np.testing.assert_allclose(out["obs"], d["obs"], rtol=OBS_RTOL, atol=OBS_ATOL)
    assert out["obs"].shape[1] == 17 and set(np.unique(out["obs"][:, 16])) <= {0.0, 1.0}
    assert np.abs(out["reward"] - d["reward"]).max() < REWARD_ATOL, np.abs(out["reward"] - d["reward"]).max()


def test_v1_reset_distribution_vs_reference():
    d = G.load("v1_reset_samples")
    n = 20000
    env = O.OracleEnv(O.reference_quad_config(num_envs=n, seed=77, variant=O.TASK_V1_SCALED17))
    obs = env.reset()
    K = (env.istate[O.I_FLAGS] >> 4) & 15
    assert set(np.unique(K)) == {1, 2} and abs((K == 2).mean() - 0.5) < 0.02 and abs((d["K"] == 2).mean() - 0.5) < 0.02
    st, wp0 = env.fstate[0:3].T, env.fstate[O.F_WP0:O.F_WP0 + 3].T
    assert (np.abs(st[:, :2]) <= 1).all() and (st[:, 2] >= 1).all() and (st[:, 2] < 2).all()
    assert (wp0[:, 2] >= 1).all() and (wp0[:, 2] <= 3).all() and (np.abs(wp0[:, :2]) <= 1).all()
    from scipy import stats
    for c in range(3):
        assert stats.ks_2samp(st[:, c], d["start"][:, c]).pvalue > 1e-3
        assert stats.ks_2samp(wp0[:, c], d["waypoints"][:, 0, c]).pvalue > 1e-3
    two = K == 2
    wp1 = env.fstate[O.F_WP0 + 3:O.F_WP0 + 6].T[two]
    for c in range(3):
        assert stats.ks_2samp(wp1[:, c], d["waypoints"][d["K"] == 2][:, 1, c]).pvalue > 1e-3
    assert (obs[:, 16] == (K == 1)).all()                       # is_final flag of the first observation
    assert (env.fstate[O.F_WP0 + 3:O.F_WP0 + 6].T[~two] == 0).all()
