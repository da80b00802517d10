"""BASELINE config 3 on the GPU: hexacopter + 3-joint arm.  No reference dynamics exist for this vehicle (parity
unpinned): the HIP kernel is checked against the fp64 oracle of the same specified model (which tests/test_arm_cpu.py pins
by physical invariants), and directly against momentum conservation."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from tests.test_arm_cpu import arm_cfg, momenta
from tests.test_gpu_parity import gpu_state, rel_err, REL32, OBS_ULP

pytestmark = pytest.mark.gpu


def orc_arm(n, seed=0, flags=O.FLAG_AUTO_RESET, **over):
    cfg = arm_cfg(**over)
    cfg.num_envs = n; cfg.seed = seed; cfg.flags = flags
    return O.OracleEnv(cfg)


def both_step(env, orc, a):
    import torch
    f, i = gpu_state(env)
    orc.fstate[:] = f; orc.istate[:] = i
    obs, rew, done, info = env.step(torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda())
    torch.cuda.synchronize()
    g = dict(obs=obs.cpu().numpy().copy(), reward=rew.cpu().numpy().astype(np.float64), done=done.cpu().numpy().copy(),
             info=info.cpu().numpy().view(np.uint32).copy())
    return g, orc.step(a)


def rand_actions(rng, n, scale=0.05):
    a = rng.uniform([0.6, -1, -1, -1, -1, -1, -1], [1.4, 1, 1, 1, 1, 1, 1], (n, 7)).astype(np.float32)
    a[:, 1:4] *= scale
    return a


def test_team_fp64_build_rk4_substeps_and_ragged_batches():
    """The fp64 lane-team build with RK4 sub-steps and batches that do not fill a tile: <= 1e-12 against the oracle, reset states bit-exact;
    amenv_rollout refuses it (the build is a logic gate of amenv_step)."""
    import torch
    import rl_aerial_manipulator_amd as amd
    for n, sub in ((1, 1), (67, 2), (300, 3)):
        rng = np.random.RandomState(n)
        env = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=4, dtype="f64", kernel="team", rk4_substeps=sub, max_episode_steps=40)
        assert "step_kernel_team<double" in env.kernel_name
        orc = orc_arm(n, seed=4)
        orc.cfg.task.rk4_substeps = sub; orc.cfg.task.max_episode_steps = 40
        env.reset()
        f, i = gpu_state(env)
        q = rng.normal(size=(4, n)) * 0.2; q[0] += 1; q /= np.linalg.norm(q, axis=0)
        f[6:10] = q; f[10:13] = rng.normal(0, 1.0, (3, n)); f[19:22] = rng.uniform(-1, 1, (3, n)); f[22:25] = rng.normal(0, 1.0, (3, n))
        env.set_state(f, i)
        worst = 0.0
        for t in range(60):
            g, o = both_step(env, orc, rand_actions(rng, n))
            f2, i2 = gpu_state(env)
            assert np.array_equal(g["info"] & 127, o["info"] & 127) and np.array_equal(i2, orc.istate)
            nd = o["done"] == 0
            worst = max(worst, rel_err(f2[:, nd], orc.fstate[:, nd]).max() if nd.any() else 0.0)
            assert np.array_equal(f2[:, ~nd], orc.fstate[:, ~nd])
            np.testing.assert_allclose(g["reward"], o["reward"], rtol=1e-9, atol=1e-9)
        assert worst < 1e-12, (n, sub, worst)
        with pytest.raises(amd.AmenvError, match="logic gate"):
            env.rollout(torch.zeros(2, n, 7, device="cuda"))
        env.close()


def test_arm_fp64_build_exists_and_uses_the_lane_kernel():
    """the fp64 logic-check build of the arm kernel (looped RK4: one copy of the RHS, no scratch)."""
    import rl_aerial_manipulator_amd as amd
    env = amd.GpuWaypointEnv(64, vehicle="hexa_arm", dtype="f64")
    assert "step_kernel<double" in env.kernel_name and "arm3" in env.kernel_name
    env.close()
    with pytest.raises(amd.AmenvError, match="AMENV_KERNEL_HELPER"):
        amd.GpuWaypointEnv(64, vehicle="hexa_arm", dtype="f64", kernel="helper")   # the two-wave kernel is fp32 only


def test_arm_dims_and_reset():
    import rl_aerial_manipulator_amd as amd
    n = 1000
    env = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=5)
    assert (env.obs_dim, env.act_dim, env.n_float_fields) == (29, 7, 25) and "arm" in env.kernel_name
    orc = orc_arm(n, seed=5)
    obs = env.reset().cpu().numpy(); oobs = orc.reset()
    f, i = gpu_state(env)
    assert np.array_equal(f, orc.fstate) and np.array_equal(i, orc.istate) and (f[19:25] == 0).all()
    # obs[13:16] = (waypoint - (base + tool offset)) / 2: the fp32 sum base + offset rounds once more than the rigid observation
    np.testing.assert_allclose(obs, oobs, rtol=OBS_ULP, atol=1.5e-7)
    # at home and level the tool point hangs at a constant offset below the base (forward kinematics with all joints at 0)
    home = np.array(orc.cfg.vehicle.joint_origin[:9]).reshape(3, 3).sum(0) + np.array(orc.cfg.vehicle.tool_offset[:3])
    np.testing.assert_allclose(obs[:, 26:29], np.tile(home / 0.5, (n, 1)), rtol=2e-7)
    np.testing.assert_allclose(env.ee_position().cpu().numpy(), orc.ee_position(), rtol=0, atol=3e-7)
    assert env.bytes_per_env_step == 4 * (13 + 3 + 3 + 6) + 12 + 28 + 4 * (15 + 6) + 12 + 116 + 4 + 1 + 4
    env.close()


# fp32 vs fp64 oracle: rounding level, far inside the 1e-5 gate; fp64 build: the logic gate; "team" = the lane-team kernel (16 lanes per env),
# "staged" = the stage-wave kernel (four RK4 stage waves + main wave per tile, base dynamics on joint-configuration aggregates)
@pytest.mark.parametrize("dtype,tol,kernel", [("f32", 2e-6, "lane"), ("f32", 2e-6, "helper"), ("f32", 3e-6, "team"), ("f32", 3e-6, "staged"), ("f64", 1e-12, "lane"),
                                               ("f64", 1e-12, "team"),     # fp64 build of the lane-team kernel: the logic gate of its DPP plumbing (selectors, row sums, stage hand-over)
                                               ("f64", 1e-12, "staged")])  # fp64 build of the stage-wave kernel: the logic gate of its LDS hand-over (stage -> slot map, joint integration on wave 3)
def test_arm_closed_loop_vs_oracle(dtype, tol, kernel):
    """Teacher-forced per step (the oracle is re-seated on the GPU state each step), joints slewing, with resets."""
    import rl_aerial_manipulator_amd as amd
    n = 2048
    rng = np.random.RandomState(3)
    env = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=9, dtype=dtype, max_episode_steps=120, kernel=kernel)
    assert {"lane": "step_kernel<", "helper": "arm2w", "team": "step_kernel_team", "staged": "step_kernel_armk"}[kernel] in env.kernel_name
    orc = orc_arm(n, seed=9)
    orc.cfg.task.max_episode_steps = 120
    env.reset()
    # start from disturbed attitudes / rates / joint states so every coupling term is exercised
    f, i = gpu_state(env)
    q = rng.normal(size=(4, n)) * 0.2; q[0] += 1; q /= np.linalg.norm(q, axis=0)
    f[6:10] = q; f[10:13] = rng.normal(0, 1.0, (3, n)); f[3:6] = rng.normal(0, 0.5, (3, n))
    f[19:22] = rng.uniform(-1, 1, (3, n)); f[22:25] = rng.normal(0, 1.0, (3, n))
    env.set_state(f if dtype == "f64" else f.astype(np.float32), i)
    worst = 0.0; flips = 0; dones = 0; worst_obs = 0.0
    for t in range(150):
        a = rand_actions(rng, n)
        a[::9, 0] = 0.0
        g, o = both_step(env, orc, a)
        f2, i2 = gpu_state(env)
        bad = np.nonzero((g["info"] & 127) != (o["info"] & 127))[0]
        flips += len(bad)
        ok = np.ones(n, bool); ok[bad] = False
        nd = ok & (o["done"] == 0)
        rows = np.r_[0:15, 16:25]   # everything but the running episode return (row 15: reward-threshold flips accumulate there)
        worst = max(worst, rel_err(f2[rows][:, nd], orc.fstate[rows][:, nd]).max())
        worst_obs = max(worst_obs, rel_err(g["obs"][ok], o["obs"][ok]).max())
        dn = ok & (o["done"] != 0)
        dones += int(dn.sum())
        assert np.array_equal(f2[:, dn], orc.fstate[:, dn]) and np.array_equal(i2[:, ok], orc.istate[:, ok])
    from tests.golden_util import report_flips
    report_flips(f"test_arm_closed_loop_vs_oracle[{dtype}-{kernel}] (flag bits vs the oracle, env-steps)", flips, 150 * n)
    assert worst < tol, worst
    assert worst_obs < max(tol, 1.3e-7) and flips <= 4 and dones > 100, f"obs {worst_obs}, {flips} threshold flips in {150 * n} env-steps, {dones} episode ends"
    env.close()


def test_arm_momentum_conservation_on_gpu():
    """g = 0, zero rotor wrench, joints slewing: total linear / angular momentum of the GPU trajectory stay constant
    (fp64 build: integration error only; fp32 build: rounding-limited)."""
    import torch
    import rl_aerial_manipulator_amd as amd
    n = 64
    for dtype, tol in (("f32", 2e-4), ("f64", 2e-7)):   # rounding-limited in fp32; fp64 build: RK4 truncation error only (measured 3e-8 over these 300 steps)
        cfg = amd._lib.default_config("hexa_arm", n)
        cfg.vehicle.g = 0.0
        for r in range(8):
            cfg.vehicle.t_min[r] = 0.0
        cfg.dtype = amd._lib.F64 if dtype == "f64" else amd._lib.F32
        cfg.flags = 0
        env = amd.GpuWaypointEnv(n, config=cfg)
        env.reset()
        rng = np.random.RandomState(0)
        f, i = gpu_state(env)
        q = rng.normal(size=(4, n)); q /= np.linalg.norm(q, axis=0)
        f[6:10] = q; f[3:6] = rng.normal(0, 0.3, (3, n)); f[10:13] = rng.normal(0, 0.7, (3, n)); f[19:22] = rng.uniform(-0.5, 0.5, (3, n))
        env.set_state(f if dtype == "f64" else f.astype(np.float32), i)
        f, _ = gpu_state(env)
        ocfg = arm_cfg(g=0.0)
        s_of = lambda F, k: np.concatenate([F[0:13, k], F[19:25, k]])
        P0 = [momenta(ocfg, s_of(f, k)) for k in range(n)]
        a = np.zeros((n, 7), np.float32); a[:, 4:] = rng.uniform(-1, 1, (n, 3))
        at = torch.from_numpy(a).cuda()
        for t in range(300):
            env.step(at)
        f, _ = gpu_state(env)
        worst = 0.0
        for k in range(n):
            P, L, _ = momenta(ocfg, s_of(f, k))
            worst = max(worst, np.abs(P - P0[k][0]).max() / max(1e-3, np.abs(P0[k][0]).max()), np.abs(L - P0[k][1]).max() / max(1e-3, np.abs(P0[k][1]).max()))
        assert np.abs(f[22:25]).max() < 5 and np.abs(f[19:22] - 0).max() > 0.3
        assert worst < tol, (dtype, worst)
        env.close()


def test_arm_rollout_equals_steps_and_vecenv():
    import torch
    import rl_aerial_manipulator_amd as amd
    n, T = 300, 40
    rng = np.random.RandomState(2)
    a = np.stack([rand_actions(rng, n) for _ in range(T)])
    a[:, ::7, 0] = 0.0
    at = torch.from_numpy(a).cuda()
    e1 = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=3, max_episode_steps=30); e2 = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=3, max_episode_steps=30)
    e1.reset(); e2.reset()
    ro = e1.rollout(at)
    assert ro["obs"].shape == (T, n, 29)
    for t in range(T):
        obs, rew, done, info = e2.step(at[t])
        assert torch.equal(ro["obs"][t], obs) and torch.equal(ro["reward"][t], rew) and torch.equal(ro["info_bits"][t], info)
    assert ro["done"].sum().item() > 0
    ve = amd.GpuVecEnv(num_envs=64, vehicle="hexa_arm")
    assert ve.observation_space.shape == (29,) and ve.action_space.shape == (7,)
    assert ve.reset().shape == (64, 29)
    o, r, d, inf = ve.step(np.tile(np.array([1, 0, 0, 0, 0.2, -0.2, 0.1], np.float32), (64, 1)))
    assert o.shape == (64, 29) and np.isfinite(o).all()
    e1.close(); e2.close(); ve.close()


def test_two_wave_kernel_is_bit_identical_to_one_wave_kernel():
    """Small batches run step_kernel_arm2w (main + helper wave per 64-env tile, link 3 on the helper); its partial sums are
    added in the one-wave kernel's order, so whole trajectories -- resets, rewards, observations included -- are identical."""
    import torch
    import rl_aerial_manipulator_amd as amd
    outs = []
    for kern in ("lane", "helper"):               # amenv_config.step_kernel: AMENV_KERNEL_LANE / AMENV_KERNEL_HELPER
        env = amd.GpuWaypointEnv(300, vehicle="hexa_arm", seed=9, kernel=kern)
        assert ("arm2w" in env.kernel_name) == (kern == "helper")
        env.reset()
        g = torch.Generator(device="cuda").manual_seed(5)
        acts = torch.randn(400, 300, 7, device="cuda", generator=g) * 0.3
        acts[..., 0] += 1.0
        acts[:, ::7, 0] = 0.2                      # some envs fall: crashes and auto-resets inside the window
        acts = acts.clamp(-1, 2)
        rec = []
        for t in range(400):
            obs, rew, done, info = env.step(acts[t])
            d = done.bool()
            rec.append((obs.clone(), rew.clone(), done.clone(), info.clone(), env.terminal_obs[d].clone(), env.ep_return[d].clone(), env.ep_len[d].clone()))
        f, i = env.get_state()
        outs.append((rec, f.clone(), i.clone(), env.stats()))
        env.close()
    (r0, f0, i0, s0), (r1, f1, i1, s1) = outs
    assert torch.equal(f0, f1) and torch.equal(i0, i1) and s0 == s1 and s0["episodes"] > 20
    n_term = 0
    for (o0, w0, d0, b0, t0, er0, el0), (o1, w1, d1, b1, t1, er1, el1) in zip(r0, r1):
        assert torch.equal(o0, o1) and torch.equal(w0, w1) and torch.equal(d0, d1) and torch.equal(b0, b1)
        assert torch.equal(t0, t1) and torch.equal(er0, er1) and torch.equal(el0, el1)      # SB3 terminal_observation / Monitor values
        n_term += len(t0)
    assert n_term > 20


@pytest.mark.parametrize("n", [1, 63, 65, 129])
def test_two_wave_kernel_ragged_batches(n):
    """Batches that do not fill their last tile: the two kernels still agree bit for bit and nothing is written past row n."""
    import torch
    import rl_aerial_manipulator_amd as amd
    outs = []
    for kern in ("lane", "helper"):
        env = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=2, kernel=kern)
        env.reset()
        g = torch.Generator(device="cuda").manual_seed(1)
        acts = (torch.randn(60, n, 7, device="cuda", generator=g) * 0.3)
        acts[..., 0] += 0.6                       # sinking: crashes and resets within the window
        acts = acts.clamp(-1, 2)
        for t in range(60):
            obs, rew, done, info = env.step(acts[t])
        f, i = env.get_state()
        outs.append((obs.clone(), rew.clone(), f.clone(), i.clone(), env.stats()))
        env.close()
    a, b = outs
    assert all(torch.equal(x, y) for x, y in zip(a[:4], b[:4])) and a[4] == b[4]
    assert bool(torch.isfinite(a[0]).all())


@pytest.mark.parametrize("vehicle,n,kernel,name", [("hexa_arm", 4096, "auto", "step_kernel_team"), ("hexa_arm", 16384, "auto", "step_kernel_armk"),
                                                   ("hexa", 4096, "team", "step_kernel_quad")])
def test_arm_long_run_stays_finite(vehicle, n, kernel, name):
    """1e8 env-steps with random actions on the headline configuration (lane-team kernel), on the stage-wave kernel and on the lane-quad
    rigid kernel: no non-finite state, every episode ends by crash, bounds or time."""
    import torch
    import rl_aerial_manipulator_amd as amd
    env = amd.GpuWaypointEnv(n, vehicle=vehicle, seed=1, kernel=kernel)
    assert name in env.kernel_name
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(3)
    ring = torch.randn(64, n, env.act_dim, device="cuda", generator=g) * 0.5
    ring[..., 0] += 1.0
    ring = ring.clamp(-1, 2).contiguous()
    graph = torch.cuda.CUDAGraph()
    for t in range(64):
        env.step(ring[t])
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        for t in range(64):
            env.step(ring[t])
    for _ in range(100_000_000 // n // 64):
        graph.replay()
    torch.cuda.synchronize()
    s = env.stats()
    f, _ = env.get_state()
    assert s["nonfinite"] == 0 and bool(torch.isfinite(f).all()) and bool(torch.isfinite(env.obs).all())
    assert s["episodes"] > 10000 and s["episodes"] == s["terminated"] + s["truncated"]


@pytest.mark.parametrize("kernel", ["lane", "team", "staged"])
@pytest.mark.parametrize("ee_task", ["tool", "base"])
def test_forward_kinematics_and_tool_point_task_vs_oracle(ee_task, kernel):
    """amenv_ee_position / obs[26:29] (forward kinematics) and the task point of the reward / reach test against the fp64 oracle, on
    random attitudes and joint angles, for both task modes; a third of the envs sit with the TASK point inside the waypoint ball."""
    import rl_aerial_manipulator_amd as amd
    n = 4096
    rng = np.random.RandomState(11)
    env = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=4, ee_task=ee_task, kernel=kernel)
    orc = orc_arm(n, seed=4)
    orc.cfg.task.ee_task = O.EE_TASK_TOOL if ee_task == "tool" else O.EE_TASK_BASE
    env.reset()
    f, i = gpu_state(env)
    q = rng.normal(size=(4, n)); q[0] += 2; q /= np.linalg.norm(q, axis=0)
    f[6:10] = q
    f[19:22] = rng.uniform([[-3.1], [-1.5], [-1.5]], [[3.1], [1.5], [1.5]], (3, n))
    f[22:25] = rng.normal(0, 0.5, (3, n)); f[10:13] = rng.normal(0, 0.3, (3, n))
    env.set_state(f.astype(np.float32), i)
    f, i = gpu_state(env)
    orc.fstate[:] = f; orc.istate[:] = i
    ee_g = env.ee_position().cpu().numpy(); ee_o = orc.ee_position()
    assert np.abs(ee_g - ee_o).max() < 2e-6
    assert np.linalg.norm(ee_o - f[0:3].T, axis=1).min() > 0.05 and np.linalg.norm(ee_o - f[0:3].T, axis=1).max() < 0.4
    assert rel_err(env.observe().cpu().numpy(), orc.observe()).max() < REL32
    # put the waypoint next to the task point of a third of the envs (the other point is ~0.3 m away: outside the ball)
    sel = rng.rand(n) < 0.33
    tp = ee_o if ee_task == "tool" else f[0:3].T
    f[O.F_WP0:O.F_WP0 + 3, sel] = (tp[sel] + rng.normal(0, 0.03, (sel.sum(), 3))).T
    env.set_state(f.astype(np.float32), i)
    a = rand_actions(rng, n)
    g, o = both_step(env, orc, a)
    bad = (g["info"] & 127) != (o["info"] & 127)
    assert bad.sum() <= 6, bad.sum()
    ok = ~bad
    assert ((o["info"][ok] & O.INFO_SUCCESS) != 0).sum() > n // 5
    assert rel_err(g["obs"][ok], o["obs"][ok]).max() < REL32
    assert (np.abs(g["reward"] - o["reward"]) / np.maximum(1, np.abs(o["reward"])))[ok].max() < 5e-5
    env.close()


@pytest.mark.parametrize("n,kernel", [(1, "team"), (3, "team"), (63, "team"), (65, "team"), (300, "team"), (4096, "team"),
                                      (1, "staged"), (65, "staged"), (300, "staged"), (8192, "staged")])
def test_team_kernel_tracks_the_lane_kernel(n, kernel):
    """The lane-team kernel (AMENV_KERNEL_TEAM: 16 lanes per env, DPP exchanges) and the stage-wave kernel (AMENV_KERNEL_STAGED: the RK4
    stages' joint-configuration sums on four wavefronts, base dynamics on the aggregates) against the one-lane-per-env kernel on the same inputs:
    same resets (bit-exact: same Philox words, same fp32 arithmetic), same flags, states within fp32 rounding after every step
    (sums are associated differently, so not bit for bit), Monitor totals equal; ragged batches write nothing past row n."""
    import torch
    import rl_aerial_manipulator_amd as amd
    T = 150
    g = torch.Generator(device="cuda").manual_seed(7)
    acts = torch.randn(T, n, 7, device="cuda", generator=g) * 0.2
    acts[..., 0] += 1.0
    acts[:, ::3, 0] = 0.15                      # a third of the envs sink: crashes and auto-resets inside the window
    acts = acts.clamp(-1, 2)
    lane = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=6, kernel="lane", max_episode_steps=90)
    team = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=6, kernel=kernel, max_episode_steps=90)
    assert {"team": "step_kernel_team", "staged": "step_kernel_armk"}[kernel] in team.kernel_name
    o0 = lane.reset().clone(); o1 = team.reset().clone()
    assert torch.equal(o0, o1)
    guard = torch.full((n + 8, 29), 7.0, device="cuda")       # the team env writes its observations into rows [0, n) of this buffer only
    worst = 0.0; flips = 0; dones = 0; bonus_flips = 0
    for t in range(T):
        f, i = lane.get_state()
        team.set_state(f, i)                                   # teacher-forced: both kernels step the same state
        ol, rl, dl, il = (x.clone() for x in lane.step(acts[t]))
        ot, rt, dt, it = team.step_into(acts[t], guard[:n], team.reward, team.done)
        same = (il & 127) == (it & 127)
        flips += int((~same).sum())
        f1, i1 = lane.get_state(); f2, i2 = team.get_state()
        nd = same & (dl == 0)
        if bool(nd.any()):
            worst = max(worst, float(((f1 - f2).abs() / f1.abs().clamp(min=1.0))[:15][:, nd].max()), float(((f1 - f2).abs() / f1.abs().clamp(min=1.0))[16:][:, nd].max()))
            assert float((ol[nd] - ot[nd]).abs().max()) < 1e-5
            # rewards agree to rounding except on an env sitting exactly on the `progress > 0` threshold of the +2 bonus
            # (v2/rl_env_scaledObs.py:214-218): there the two kernels may differ by exactly that bonus
            dr = (rl - rt).abs()[nd]
            off = dr / rl.abs().clamp(min=1.0)[nd] >= 1e-4
            bonus_flips += int(off.sum())
            assert bool(((dr[off] - 2.0).abs() < 1e-3).all()), dr[off]
        dn = same & (dl != 0)
        dones += int(dn.sum())
        assert torch.equal(f1[:, dn], f2[:, dn]) and torch.equal(i1[:, same], i2[:, same]) and torch.equal(ol[dn], ot[dn])   # reset states / observations: bit-exact
        if bool(dn.any()):
            assert torch.equal(lane.ep_len[dn], team.ep_len[dn]) and float((lane.terminal_obs[dn] - team.terminal_obs[dn]).abs().max()) < 1e-5
    assert worst < 2e-6 and flips <= max(2, n // 500) and dones >= n // 4 and bonus_flips <= 1e-4 * n * T + 1, (worst, flips, dones, bonus_flips)
    assert bool((guard[n:] == 7.0).all())
    sl, st = lane.stats(), team.stats()
    assert abs(sl["episodes"] - st["episodes"]) <= flips and sl["steps"] == st["steps"]
    lane.close(); team.close()


def test_team_rollout_kernel_equals_team_steps():
    """amenv_rollout on the lane-team kernel (T steps per launch, state and constants in registers between steps) against T single
    team steps: same code per step, so bit-identical outputs, state and totals; ragged batch."""
    import torch
    import rl_aerial_manipulator_amd as amd
    n, T = 1000, 120
    g = torch.Generator(device="cuda").manual_seed(11)
    acts = torch.randn(T, n, 7, device="cuda", generator=g) * 0.2
    acts[..., 0] += 1.0
    acts[:, ::3, 0] = 0.15
    acts = acts.clamp(-1, 2).contiguous()
    e1 = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=3, kernel="team", max_episode_steps=70)
    e2 = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=3, kernel="team", max_episode_steps=70)
    e1.reset(); e2.reset()
    ro = e1.rollout(acts)
    assert ro["obs"].shape == (T, n, 29)
    for t in range(T):
        obs, rew, done, info = e2.step(acts[t])
        assert torch.equal(ro["obs"][t], obs) and torch.equal(ro["reward"][t], rew) and torch.equal(ro["info_bits"][t], info) and torch.equal(ro["done"][t], done)
    f1, i1 = e1.get_state(); f2, i2 = e2.get_state()
    assert torch.equal(f1, f2) and torch.equal(i1, i2) and e1.stats() == e2.stats() and int(ro["done"].sum()) >= n
    # outputs are optional
    e1.rollout(acts[:5], want_obs=False, want_flags=False)
    e1.close(); e2.close()


@pytest.mark.parametrize("n,kernel", [(300, "team"), (4096, "team"), (6000, "auto"), (1000, "lane"), (20000, "helper"), (20000, "auto")])
def test_closed_loop_policy_rollout_kernel(n, kernel):
    """amenv_rollout_policy: T closed-loop steps in one launch (bf16-MFMA actor / critic, Gaussian sampling, clip, env step) in its two
    forms: 16 lanes per env (where amenv_step runs the lane-team kernel) and one lane per env (64 or 128 envs per workgroup).
    The env part is checked EXACTLY by replaying the recorded (clipped) actions through amenv_step on a second env (the lane form carries the
    arithmetic of the LANE / HELPER step kernels: where AUTO steps with the stage-wave kernel, the replay env is a LANE one); the policy part against
    the fp32 torch modules on the recorded observations (bf16 tolerance: means / values within 3e-2 of their scale); the samples
    statistically (z = (a - mean) / std ~ N(0, 1), log-probs consistent with the samples)."""
    import torch
    import rl_aerial_manipulator_amd as amd
    T = 96
    torch.manual_seed(5)
    pol = amd.ActorCritic(29, 7).cuda().flatten_()
    with torch.no_grad():
        pol.log_std.data.fill_(-1.2)
        for m in (pol.action_net,):                              # a less timid head than SB3's 0.01-gain init: means of O(0.3)
            m.weight.mul_(30.0)
    env = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=4, kernel=kernel, max_episode_steps=60)
    ref = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=4, kernel="lane" if "armk" in env.kernel_name else kernel, max_episode_steps=60)
    assert ("team" in env.kernel_name) == (kernel == "team" or (kernel == "auto" and n <= 6144)) and ("armk" in env.kernel_name) == (kernel == "auto" and n > 6144)
    o0 = env.reset().clone(); ref.reset()
    dev = env.device
    obs = torch.zeros(T + 1, n, 29, device=dev); acts = torch.zeros(T, n, 7, device=dev)
    logp = torch.zeros(T, n, device=dev); vals = torch.zeros(T, n, device=dev); rew = torch.zeros(T, n, device=dev)
    dones = torch.zeros(T, n, dtype=torch.uint8, device=dev); info = torch.zeros(T, n, dtype=torch.int32, device=dev)
    tobs = torch.full((T, n, 29), float("nan"), device=dev)
    env.rollout_policy(pol.flat_param, T, seed=77, draw0=5, obs=obs, actions=acts, logp=logp, values=vals, rewards=rew, dones=dones, info_bits=info, terminal_obs=tobs)
    torch.cuda.synchronize()
    assert float((obs[0] - o0).abs().max()) < 1e-6               # (forward kinematics of the home pose vs the reset path's constant offset)
    lo, hi = pol.action_low, pol.action_high
    # (1) env part: replay the clipped actions step by step on the second env -> identical observations / rewards / flags
    for t in range(T):
        o, r, d, i = ref.step(torch.max(torch.min(acts[t], hi), lo))
        assert torch.equal(o, obs[t + 1]) and torch.equal(r, rew[t]) and torch.equal(d, dones[t]) and torch.equal(i, info[t]), t
        dn = d.bool()
        if bool(dn.any()):
            assert torch.equal(ref.terminal_obs[dn], tobs[t][dn])
    f1, i1 = env.get_state(); f2, i2 = ref.get_state()
    assert torch.equal(f1, f2) and torch.equal(i1, i2) and env.stats()["episodes"] == ref.stats()["episodes"] == int(dones.sum()) > n // 2
    assert bool(torch.isnan(tobs[~dones.bool()]).all())          # terminal rows are written only where an episode ended
    # (2) policy part vs the fp32 modules on the recorded observations
    with torch.no_grad():
        flat = obs[:T].reshape(T * n, 29)
        mean32 = pol.action_net(pol.mlp_extractor.policy_net(flat)); v32 = pol.value_net(pol.mlp_extractor.value_net(flat)).reshape(-1)
    std = torch.exp(pol.log_std.detach())
    assert float((vals.reshape(-1) - v32).abs().max()) < 3e-2 * max(1.0, float(v32.abs().max()))
    z = (acts.reshape(T * n, 7) - mean32) / std                  # the noise the kernel drew, up to the bf16 error of its means
    assert abs(float(z.mean())) < 0.02 and abs(float(z.var()) - 1.0) < 0.03 and float(z.abs().max()) < 6.5
    zc = torch.corrcoef(z[: 50000].T)
    assert float((zc - torch.eye(7, device=dev)).abs().max()) < 0.03                   # independent components
    lp32 = (-0.5 * z * z - pol.log_std.detach() - 0.9189385332).sum(1)
    assert float((logp.reshape(-1) - lp32).abs().max()) < 0.5 and float((logp.reshape(-1) - lp32).abs().mean()) < 0.05
    # different draw index -> different noise; same call again from the same state -> identical (deterministic)
    env2 = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=4, kernel=kernel, max_episode_steps=60); env2.reset()
    acts2 = torch.zeros_like(acts)
    env2.rollout_policy(pol.flat_param, T, seed=77, draw0=5, obs=torch.zeros_like(obs), actions=acts2, logp=torch.zeros_like(logp), values=torch.zeros_like(vals),
                        rewards=torch.zeros_like(rew), dones=torch.zeros_like(dones))
    assert torch.equal(acts2, acts)
    env.close(); ref.close(); env2.close()


def test_closed_loop_policy_rollout_forms_draw_the_same_actions():
    """The lane-per-env form keeps the lane-team form's Philox keying, Box-Muller mapping and log-prob summation: from the same state and
    observation the first step's raw actions, log-probs and values are bit-identical (later steps differ by the env kernels' rounding)."""
    import torch
    import rl_aerial_manipulator_amd as amd
    n, T = 2048, 2
    torch.manual_seed(5)
    pol = amd.ActorCritic(29, 7).cuda().flatten_()
    with torch.no_grad():
        pol.action_net.weight.mul_(30.0)
    out = []
    for kernel in ("team", "lane"):
        env = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=4, kernel=kernel)
        env.reset()
        dev = env.device
        obs = torch.zeros(T + 1, n, 29, device=dev); acts = torch.zeros(T, n, 7, device=dev)
        logp = torch.zeros(T, n, device=dev); vals = torch.zeros(T, n, device=dev); rew = torch.zeros(T, n, device=dev)
        dones = torch.zeros(T, n, dtype=torch.uint8, device=dev)
        env.rollout_policy(pol.flat_param, T, seed=3, draw0=9, obs=obs, actions=acts, logp=logp, values=vals, rewards=rew, dones=dones)
        torch.cuda.synchronize()
        out.append((obs.clone(), acts.clone(), logp.clone(), vals.clone()))
        env.close()
    (o1, a1, l1, v1), (o2, a2, l2, v2) = out
    # rows whose entry observation is the same MLP input in both forms (the forward kinematics of the two env kernels differ in the last
    # fp32 bits; the MLP reads the observation rounded to bf16)
    same = (o1[0].bfloat16() == o2[0].bfloat16()).all(dim=1)
    assert int(same.sum()) > n // 2
    assert torch.equal(a1[0][same], a2[0][same]) and torch.equal(l1[0][same], l2[0][same]) and torch.equal(v1[0][same], v2[0][same])
    assert float((o1[1] - o2[1]).abs().max()) < 1e-4


@pytest.mark.parametrize("form", [0, 1])
def test_arm_right_hand_side_device_code_vs_oracle(form):
    """amenv_arm_rhs: the device code of both formulations of the arm vehicle's right-hand side -- form 0 = per-link Newton-Euler sums
    (lane / two-wave kernels), form 1 = joint-configuration aggregates + base dynamics on them (stage-wave / lane-team kernels) -- on
    random states, wrenches and commands: the fp64 instantiation against the fp64 oracle (logic gate, <= 1e-12), the fp32 instantiation
    to fp32 rounding.  A third of the commands saturate the joint servos."""
    import ctypes as C
    import torch
    import rl_aerial_manipulator_amd as amd
    n = 4096
    rng = np.random.RandomState(21 + form)
    s = np.zeros((n, 19))
    s[:, :6] = rng.normal(size=(n, 6)); q = rng.normal(size=(n, 4)); s[:, 6:10] = q / np.linalg.norm(q, axis=1, keepdims=True) * rng.uniform(0.98, 1.02, (n, 1))
    s[:, 10:13] = rng.normal(size=(n, 3)) * 2; s[:, 13:16] = rng.uniform(-1.5, 1.5, (n, 3)); s[:, 16:19] = rng.normal(size=(n, 3)) * 2
    w = np.concatenate([rng.uniform(5, 60, (n, 1)), rng.normal(size=(n, 3))], 1)
    cmd = s[:, 13:16] + rng.normal(size=(n, 3)) * np.where(rng.rand(n, 1) < 0.33, 2.0, 0.05)
    cfg = amd._lib.default_config("hexa_arm", 1)
    ocfg = arm_cfg()
    ref = np.stack([O.arm_rhs(ocfg, s[i], w[i, 0], w[i, 1:], cmd[i]) for i in range(n)])
    scale = np.maximum(1.0, np.abs(ref))
    for dtype, code, tol in ((torch.float64, amd._lib.F64, 1e-12), (torch.float32, amd._lib.F32, 3e-5)):
        ts, tw, tc = (torch.as_tensor(a, dtype=dtype).cuda().contiguous() for a in (s, w, cmd))
        d = torch.empty(n, 19, dtype=dtype, device="cuda")
        p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        assert amd._lib.load().amenv_arm_rhs(C.byref(cfg), form, code, p(ts), p(tw), p(tc), p(d), n, None) == 0
        err = (np.abs(d.cpu().numpy().astype(np.float64) - ref) / scale).max()
        assert err < tol, (form, dtype, err)
    # a rigid vehicle has no such right-hand side
    assert amd._lib.load().amenv_arm_rhs(C.byref(amd._lib.default_config("hexa", 1)), form, amd._lib.F64, p(ts), p(tw), p(tc), p(d), n, None) == -1


@pytest.mark.parametrize("kernel,n", [("lane", 300), ("helper", 300), ("team", 300), ("auto", 12288)])
@pytest.mark.parametrize("substeps", [2, 3])
def test_arm_rk4_substeps_vs_oracle(substeps, kernel, n):
    """amenv_task.rk4_substeps > 1 (several RK4 steps of dt / substeps per control step) on every arm kernel that carries it, teacher-forced
    against the oracle with the same setting; the stage-wave kernel is built for one sub-step: refused when asked for, not selected by AUTO."""
    import rl_aerial_manipulator_amd as amd
    rng = np.random.RandomState(5)
    env = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=3, kernel=kernel, rk4_substeps=substeps, max_episode_steps=50)
    assert "armk" not in env.kernel_name and (kernel != "auto" or "arm2w" in env.kernel_name)
    with pytest.raises(amd.AmenvError, match="AMENV_KERNEL_STAGED"):
        amd.GpuWaypointEnv(64, vehicle="hexa_arm", kernel="staged", rk4_substeps=substeps)
    orc = orc_arm(n, seed=3)
    orc.cfg.task.rk4_substeps = substeps; orc.cfg.task.max_episode_steps = 50
    env.reset()
    f, i = gpu_state(env)
    q = rng.normal(size=(4, n)) * 0.2; q[0] += 1; q /= np.linalg.norm(q, axis=0)
    f[6:10] = q; f[10:13] = rng.normal(0, 1.0, (3, n)); f[19:22] = rng.uniform(-1, 1, (3, n)); f[22:25] = rng.normal(0, 1.0, (3, n))
    env.set_state(f.astype(np.float32), i)
    worst = 0.0; flips = 0
    for t in range(40):
        g, o = both_step(env, orc, rand_actions(rng, n))
        f2, i2 = gpu_state(env)
        bad = (g["info"] & 127) != (o["info"] & 127)
        flips += int(bad.sum())
        nd = ~bad & (o["done"] == 0)
        rows = np.r_[0:15, 16:25]
        worst = max(worst, rel_err(f2[rows][:, nd], orc.fstate[rows][:, nd]).max())
    assert worst < 3e-6 and flips <= 2, (worst, flips)
    # the setting changes the trajectory (a 5 ms RK4 step vs two of 2.5 ms differ at the 1e-9 level per step: visible in fp64 only) -- check it
    # is honoured through a quantity it changes at first order: with substeps the step still advances dt, not dt / substeps
    one = amd.GpuWaypointEnv(64, vehicle="hexa_arm", seed=3, kernel="lane", dtype="f64")
    sub = amd.GpuWaypointEnv(64, vehicle="hexa_arm", seed=3, kernel="lane", dtype="f64", rk4_substeps=substeps)
    a = np.tile(np.array([[1.3, 0, 0, 0, 0.5, 0.5, 0.5]], np.float32), (64, 1))
    import torch
    one.reset(); sub.reset()
    for _ in range(20):
        one.step(torch.from_numpy(a).cuda()); sub.step(torch.from_numpy(a).cuda())
    f1, _ = gpu_state(one); f2_, _ = gpu_state(sub)
    d = np.abs(f1[:13] - f2_[:13]).max()
    assert 0 < d < 1e-6, d          # same motion to integration error, not bit for bit


@pytest.mark.parametrize("nj,dtype,kernel,tol", [(1, "f64", "lane", 1e-12), (2, "f64", "lane", 1e-12), (2, "f64", "team", 1e-12), (1, "f32", "team", 3e-6), (2, "f32", "auto", 3e-6),
                                                 (2, "f32", "staged", 3e-6), (1, "f32", "helper", 3e-6)])
def test_n_link_arm_with_fewer_joints(nj, dtype, kernel, tol):
    """north_star's "n-link arm": amenv_vehicle.n_joints = 1 or 2.  Inside, the vehicle is the 3-joint one with phantom links behind the real
    ones (zero mass / inertia / offset: exact zeros in every sum), so EVERY kernel family serves it; the C ABI keeps the caller's dimensions
    (4 + n actions, 20 + 2 n + 3 observations, 2 n joint state fields).  Teacher-forced per step against the oracle, which runs the SHORTER chain
    natively: fp64 builds <= 1e-12 (the phantom links add nothing), fp32 to rounding; reset states bit-exact; amenv_rollout refuses."""
    import torch
    import rl_aerial_manipulator_amd as amd
    n = 777
    rng = np.random.RandomState(10 * nj + len(kernel))
    lm = [0.082, 0.054, 0.220]
    mass = 3.2121 - sum(lm[nj:])
    cfg = amd._lib.default_config("hexa_arm", n)
    cfg.vehicle.n_joints = nj; cfg.vehicle.mass = mass
    cfg.seed = 9; cfg.task.max_episode_steps = 100
    cfg.dtype = amd._lib.F64 if dtype == "f64" else amd._lib.F32
    cfg.step_kernel = amd._lib.KERNELS[kernel]
    env = amd.GpuWaypointEnv(n, config=cfg)
    assert (env.obs_dim, env.act_dim, env.n_float_fields) == (23 + 2 * nj, 4 + nj, 19 + 2 * nj) and f"{nj}-joint arm" in env.kernel_name
    orc = orc_arm(n, seed=9, n_joints=nj, mass=mass)
    orc.cfg.task.max_episode_steps = 100
    obs0 = env.reset().cpu().numpy(); oobs0 = orc.reset()
    f, i = gpu_state(env)
    assert np.array_equal(f, orc.fstate) and np.array_equal(i, orc.istate)
    np.testing.assert_allclose(obs0, oobs0, rtol=OBS_ULP, atol=1.5e-7)
    np.testing.assert_allclose(env.observe().cpu().numpy(), oobs0, rtol=OBS_ULP, atol=1.5e-7)
    np.testing.assert_allclose(env.ee_position().cpu().numpy(), orc.ee_position(), rtol=0, atol=3e-7)
    q = rng.normal(size=(4, n)) * 0.2; q[0] += 1; q /= np.linalg.norm(q, axis=0)
    f[6:10] = q; f[10:13] = rng.normal(0, 1.0, (3, n)); f[3:6] = rng.normal(0, 0.5, (3, n))
    f[19:19 + nj] = rng.uniform(-1, 1, (nj, n)); f[19 + nj:19 + 2 * nj] = rng.normal(0, 1.0, (nj, n))
    env.set_state(f if dtype == "f64" else f.astype(np.float32), i)
    worst = 0.0; worst_obs = 0.0; flips = 0; dones = 0
    for t in range(130):
        a = rng.uniform([0.6] + [-1] * (3 + nj), [1.4] + [1] * (3 + nj), (n, 4 + nj)).astype(np.float32)
        a[:, 1:4] *= 0.05
        a[::9, 0] = 0.0
        g, o = both_step(env, orc, a)
        f2, i2 = gpu_state(env)
        bad = np.nonzero((g["info"] & 127) != (o["info"] & 127))[0]
        flips += len(bad)
        ok = np.ones(n, bool); ok[bad] = False
        nd = ok & (o["done"] == 0)
        rows = np.r_[0:15, 16:19 + 2 * nj]
        if nd.any():            # (at the time limit every env that has not crashed yet ends in the same step)
            worst = max(worst, rel_err(f2[rows][:, nd], orc.fstate[rows][:, nd]).max())
        worst_obs = max(worst_obs, rel_err(g["obs"][ok], o["obs"][ok]).max())
        dn = ok & (o["done"] != 0)
        dones += int(dn.sum())
        assert np.array_equal(f2[:, dn], orc.fstate[:, dn]) and np.array_equal(i2[:, ok], orc.istate[:, ok])
        if dn.any():
            tob = env.terminal_obs.cpu().numpy()
            assert np.isfinite(tob[dn]).all() and tob.shape[1] == 23 + 2 * nj
    assert worst < tol and worst_obs < max(tol, 1.3e-7) and flips <= 3 and dones > 50, (worst, worst_obs, flips, dones)
    with pytest.raises(amd.AmenvError, match="1 or 2 joints|logic gate"):
        env.rollout(torch.zeros(2, n, 4 + nj, device="cuda"))
    env.close()


@pytest.mark.parametrize("K,dtype,tol", [(3, "f64", 1e-12), (2, "f32", 3e-6), (4, "f32", 3e-6)])
def test_arm_with_several_waypoints(K, dtype, tol):
    """The arm vehicle on the multi-waypoint v2 task (2..4 waypoints per episode; the lane kernel: the joint groups sit behind K waypoint groups):
    teacher-forced per step against the oracle with waypoint switching (+100, next-waypoint entries of the observation), resets bit-exact; the
    one-waypoint kernel families refuse the combination."""
    import torch
    import rl_aerial_manipulator_amd as amd
    n = 600
    rng = np.random.RandomState(K)
    env = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=12, dtype=dtype, num_waypoints=K, max_episode_steps=120)
    assert "step_kernel<" in env.kernel_name and f"KW={amd._lib.MAX_WAYPOINTS}" in env.kernel_name and env.n_float_fields == 16 + 3 * K + 6
    cfg = arm_cfg()
    multi = O.reference_quad_config(n, seed=12, num_waypoints=K)
    C.memmove(C.byref(multi.vehicle), C.byref(cfg.vehicle), C.sizeof(O.Vehicle))
    multi.task.ee_task = cfg.task.ee_task; multi.task.max_episode_steps = 120
    orc = O.OracleEnv(multi)
    obs0 = env.reset().cpu().numpy(); oobs0 = orc.reset()
    f, i = gpu_state(env)
    assert np.array_equal(f, orc.fstate) and np.array_equal(i, orc.istate)
    np.testing.assert_allclose(obs0, oobs0, rtol=OBS_ULP, atol=1.5e-7)
    # park a third of the envs ON their current waypoint (tool point = waypoint), so that the switch to the next one is exercised at once
    ee = orc.ee_position()
    park = np.arange(n) % 3 == 0
    f[0:3, park] += (f[16:19, park] - ee[park].T)
    j0 = 16 + 3 * K
    f[j0:j0 + 3] = rng.uniform(-0.5, 0.5, (3, n)); f[j0 + 3:j0 + 6] = rng.normal(0, 0.5, (3, n))
    f[0:3, ~park] += rng.normal(0, 0.05, (3, int((~park).sum())))
    env.set_state(f if dtype == "f64" else f.astype(np.float32), i)
    worst = 0.0; worst_obs = 0.0; flips = 0; switched = 0; dones = 0
    for t in range(140):
        a = rand_actions(rng, n)
        a[::9, 0] = 0.0
        g, o = both_step(env, orc, a)
        f2, i2 = gpu_state(env)
        bad = np.nonzero((g["info"] & 127) != (o["info"] & 127))[0]
        flips += len(bad)
        ok = np.ones(n, bool); ok[bad] = False
        nd = ok & (o["done"] == 0)
        rows = np.r_[0:15, 16:16 + 3 * K + 6]
        if nd.any():
            worst = max(worst, rel_err(f2[rows][:, nd], orc.fstate[rows][:, nd]).max())
        worst_obs = max(worst_obs, rel_err(g["obs"][ok], o["obs"][ok]).max())
        switched += int(((i2[O.I_FLAGS] & 15) > 0)[ok].sum())
        dn = ok & (o["done"] != 0)
        dones += int(dn.sum())
        assert np.array_equal(f2[:, dn], orc.fstate[:, dn]) and np.array_equal(i2[:, ok], orc.istate[:, ok])
    assert worst < tol and worst_obs < max(tol, 1.3e-7) and flips <= 3 and switched > 1000 and dones > 50, (worst, worst_obs, flips, switched, dones)
    for kernel in ("team", "staged", "helper"):
        with pytest.raises(amd.AmenvError):
            amd.GpuWaypointEnv(64, vehicle="hexa_arm", num_waypoints=K, kernel=kernel)
    env.close()


@pytest.mark.parametrize("kernel", ["team", "staged", "lane"])
def test_arm_kernels_with_nan_guard_and_without_auto_reset(kernel):
    """The configuration switches on the arm kernels (the lane-team kernel's helper wave decides per row whether to prepare a reset; with the NaN
    guard it always does): (1) AMENV_FLAG_NAN_GUARD: a poisoned env ends with terminated | NONFINITE, reward -100, and is reset; the others
    step exactly as without the flag; (2) auto-reset off: an env that ends is NOT reset (state keeps integrating, flags say done), Monitor
    outputs and totals still count it."""
    import torch
    import rl_aerial_manipulator_amd as amd
    n = 333
    act = torch.tensor([[1.0, 0.02, -0.01, 0.0, 0.3, -0.2, 0.1]], device="cuda").repeat(n, 1)
    plain = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=3, kernel=kernel)
    guard = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=3, kernel=kernel, nan_guard=True)
    plain.reset(); guard.reset()
    for _ in range(3):
        o1, r1, d1, i1 = plain.step(act); o2, r2, d2, i2 = guard.step(act)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(i1, i2)
    f, i = guard.get_state()
    f[3, 7] = float("nan"); f[1, 200] = float("inf")
    guard.set_state(f, i)
    o2, r2, d2, i2 = guard.step(act)
    bits = i2.cpu().numpy().view(np.uint32)
    assert (bits[[7, 200]] & O.INFO_NONFINITE).all() and (bits[[7, 200]] & O.INFO_WAS_RESET).all() and d2[7] == 1 and d2[200] == 1 and float(r2[7]) == -100.0
    assert (np.delete(bits, [7, 200]) & O.INFO_NONFINITE == 0).all() and bool(torch.isfinite(o2).all())
    f2, i2s = guard.get_state()
    assert int(i2s[3, 7]) == 2 and int(i2s[0, 7]) == 0 and bool(torch.isfinite(f2).all())
    plain.close(); guard.close()
    # (2) no auto-reset: drop env 11 to the ground
    env = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=3, kernel=kernel, auto_reset=False)
    env.reset(); env.stats(reset=True)
    f, i = env.get_state()
    f[2, 11] = 0.05; f[5, 11] = -1.0
    env.set_state(f, i)
    o, r, d, info = env.step(act)
    bits = info.cpu().numpy().view(np.uint32)
    assert d[11] == 1 and (bits[11] & O.INFO_CRASHED) and not (bits[11] & O.INFO_WAS_RESET) and int(d.sum()) == 1
    f2, i2s = env.get_state()
    assert int(i2s[3, 11]) == 1 and int(i2s[0, 11]) == 1 and float(f2[2, 11]) < 0.05          # same episode, one step on, still falling
    st = env.stats()
    assert st["episodes"] == 1 and st["crashed"] == 1 and int(env.ep_len[11]) == 1
    env.close()
