"""Row f4 on the MI355X: the PID + minimum-snap baseline as HIP kernels (csrc/amenv_baseline.hpp) through the C ABI.
Pinned to the reference's recorded run (tests/golden/pid_helix.npz, tools/gen_golden_pid.py: `PID Controller/{pid_controller,trajGen3D}.py`
run as its runsim.py runs them), compared with the torch restatement, and flown in closed loop on the HIP environment."""
import os

import numpy as np
import pytest
import torch

import rl_aerial_manipulator_amd as amd
from rl_aerial_manipulator_amd.baselines import MinSnapTrajectory, PidController, PidWaypointPolicy

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold():
    return np.load(os.path.join(GOLD, "pid_helix.npz"))


def cu(a, dtype=torch.float64):
    return torch.as_tensor(np.asarray(a), dtype=dtype).cuda()


def test_min_snap_solve_kernel_matches_reference_mst():
    """amenv_minsnap_solve (device Gauss-Jordan of the constraint matrix + one product per trajectory) vs get_MST_coefficients."""
    g = gold()
    tr = MinSnapTrajectory(cu(g["waypoints"]), float(g["speed"]))
    assert tr.coeff.is_cuda and tr.coeff.shape == (1, 32, 3)
    for ax, name in enumerate(("coeff_x", "coeff_y", "coeff_z")):
        assert np.abs(tr.coeff[0, :, ax].cpu().numpy() - g[name]).max() < 1e-11 * max(1.0, np.abs(g[name]).max())
    # batched, other segment counts: against the CPU restatement (torch.linalg.solve of the same system)
    rng = np.random.RandomState(5)
    for n in (1, 2, 7, 16):
        w = rng.uniform(-3, 3, size=(33, n + 1, 3))
        a, b = MinSnapTrajectory(cu(w), 0.9), MinSnapTrajectory(torch.from_numpy(w), 0.9)
        scale = float(b.coeff.abs().max())
        assert float((a.coeff.cpu() - b.coeff).abs().max()) < 1e-9 * scale, n
        assert torch.allclose(a.T.cpu(), b.T, rtol=1e-14, atol=0) and torch.allclose(a.S.cpu(), b.S, rtol=1e-14, atol=1e-15)


def test_min_snap_eval_kernel_matches_reference_generate_trajectory():
    g = gold()
    tr = MinSnapTrajectory(cu(g["waypoints"]), float(g["speed"]))
    m = len(g["t"])
    des = tr.evaluate(cu(g["t"]), traj=torch.zeros(m, dtype=torch.int64)).cpu().numpy()       # all 1200 recorded times in ONE launch
    assert np.abs(des[:, 0:3] - g["des_pos"]).max() < 1e-11
    assert np.abs(des[:, 3:6] - g["des_vel"]).max() < 1e-11
    assert np.abs(des[:, 6:9] - g["des_acc"]).max() < 1e-10
    assert np.array_equal(des[:, 9], g["des_yaw"]) and np.array_equal(des[:, 10], g["des_yawdot"])
    assert float(g["t"][-1]) > float(tr.S[0, -1])                     # includes the hover-at-the-last-waypoint phase
    assert np.array_equal(des[0, 0:3], g["waypoints"][0]) and np.array_equal(des[-1, 0:3], g["waypoints"][-1])
    # fp32 output build, and desired_state() of a batch
    d32 = tr.evaluate(cu(g["t"]), traj=torch.zeros(m, dtype=torch.int64), dtype=torch.float32).cpu().numpy()
    assert d32.dtype == np.float32 and np.abs(d32 - des).max() < 2e-6 * max(1.0, np.abs(des).max())
    both = MinSnapTrajectory(cu(np.stack([g["waypoints"], g["waypoints"][::-1]])), float(g["speed"]))
    p2 = both.desired_state(1.7)[0].cpu().numpy()
    i17 = int(np.argmin(np.abs(g["t"] - 1.7)))
    assert np.abs(p2[0] - g["des_pos"][i17]).max() < 1e-9 and np.abs(p2[0] - p2[1]).max() > 0.1


def test_pid_kernel_matches_reference_teacher_forced():
    """amenv_pid_run, fp64: F, M and the attitude for every recorded (state, desired state) pair; the integral memory is carried from
    call to call on the device exactly as the reference's module carries it."""
    g = gold()
    pid = PidController(1, float(g["dt"]), device="cuda", dtype=torch.float64)
    state, des = cu(g["state"]), cu(np.concatenate([g["des_pos"], g["des_vel"], g["des_acc"], g["des_yaw"][:, None], g["des_yawdot"][:, None]], 1))
    Fs, Ms, Rs = [], [], []
    for i in range(len(g["t"])):
        F, M, rpy = pid.run_state(state[i:i + 1], des[i:i + 1])
        Fs.append(F); Ms.append(M); Rs.append(rpy)
    F, M, R = torch.cat(Fs).cpu().numpy(), torch.cat(Ms).cpu().numpy(), torch.cat(Rs).cpu().numpy()
    assert np.abs(R - g["rpy"]).max() < 1e-11
    assert (np.abs(F - g["F"]) / np.maximum(1.0, np.abs(g["F"]))).max() < 1e-9
    assert (np.abs(M - g["M"]).max(1) / np.maximum(1.0, np.abs(g["M"]).max(1))).max() < 1e-9
    # fp32 product build on the same pairs (integrals carried in fp32): the gains are stiff (k_p = 1000 on z), so the bound is on
    # F / (m g) and M / 0.1, the quantities that become actions
    pid32 = PidController(1, float(g["dt"]), device="cuda", dtype=torch.float32)
    e = 0.0
    for i in range(0, 400):
        F32, M32, _ = pid32.run_state(state[i:i + 1], des[i:i + 1])
        e = max(e, abs(float(F32) - g["F"][i]) / (0.18 * 9.81), float(np.abs(M32.cpu().numpy()[0] - g["M"][i]).max()) / 0.1)
    assert e < 5e-3, e


def test_pid_kernel_batched_integral_clamp_and_many_vehicles():
    n = 5000
    pid = PidController(n, 1.0, device="cuda", dtype=torch.float64)
    state = torch.zeros(n, 13, device="cuda", dtype=torch.float64)
    state[:, 6] = 1.0
    des = torch.zeros(n, 11, device="cuda", dtype=torch.float64)
    des[:, 0:3] = 1e4
    pid.run_state(state, des)
    assert float(pid.integral[:, :3].max()) == 100.0 and float(pid.integral[:, :3].min()) == 100.0     # pid_controller.py:34,66-67
    # against the torch restatement on random states
    rng = np.random.RandomState(2)
    s = rng.normal(size=(n, 13)); s[:, 6:10] /= np.linalg.norm(s[:, 6:10], axis=1, keepdims=True) * rng.uniform(0.9, 1.1, size=(n, 1))
    d = rng.normal(size=(n, 11))
    a, b = PidController(n, 0.005, device="cuda", dtype=torch.float64), PidController(n, 0.005)
    for _ in range(3):
        Fa, Ma, Ra = a.run_state(cu(s), cu(d))
        Fb, Mb, Rb = b.run_state(torch.from_numpy(s), torch.from_numpy(d))
    assert torch.allclose(Fa.cpu(), Fb, rtol=1e-12, atol=1e-12) and torch.allclose(Ma.cpu(), Mb, rtol=1e-11, atol=1e-9)
    assert torch.allclose(Ra.cpu(), Rb, rtol=0, atol=1e-12) and torch.allclose(a.integral.cpu(), b.integral, rtol=1e-13, atol=1e-15)


def test_runsim_loop_on_the_gpu_follows_the_reference_run():
    """runsim.py's loop with every piece on the MI355X: amenv_minsnap_eval -> amenv_pid_run -> amenv_step (fp64 build of the quadrotor env,
    dt = 0.01, F and M handed over as actions), free-running for the 1200 recorded steps, against the reference's own closed-loop
    trajectory.  Its Quadcopter integrates with LSODA, this env with one RK4 step of 10 ms, and the attitude loop (k_p = 160 on an
    inertia of 2.5e-4 at 100 Hz) chatters between the rotor limits, so the two runs are not the same trajectory digit for digit (the
    fp64 CPU restatement of this loop differs from the recording by up to 0.2 m as well): the test is that the GPU loop flies the
    same helix -- within the reference's own tracking error of the recorded run all the way, hovering at the last waypoint at the end."""
    g = gold()
    cfg = amd._lib.default_config("quad", 1)
    cfg.dtype, cfg.flags, cfg.task.dt = amd._lib.F64, 0, float(g["dt"])      # no auto-reset: crash / bounds flags never touch the dynamics
    env = amd.GpuWaypointEnv(1, config=cfg)
    env.reset()
    f, i = env.get_state()
    s0 = g["state"][0]
    f[:13, 0] = cu(s0)
    f[amd._lib.F_WP0:amd._lib.F_WP0 + 3, 0] = cu([50.0, 50.0, 50.0])       # a waypoint far away: the task never interferes
    env.set_state(f, i)
    tr = MinSnapTrajectory(cu(g["waypoints"]), float(g["speed"]))
    pid = PidController(1, float(g["dt"]), device="cuda", dtype=torch.float64)
    des_all = tr.evaluate(cu(g["t"]), traj=torch.zeros(len(g["t"]), dtype=torch.int64))
    err = track = 0.0
    for k in range(len(g["t"])):
        f, _ = env.get_state()
        F, M, _ = pid.run_state(f[:13, 0].unsqueeze(0), des_all[k:k + 1])
        a = torch.cat([F / (0.18 * 9.81), M[0] / 0.1]).to(torch.float32).unsqueeze(0)     # rl_env_scaledObs.py:125-126 inverted
        env.step(a)
        f, _ = env.get_state()
        pos = f[:3, 0].cpu().numpy()
        err, track = max(err, float(np.abs(pos - g["state_next"][k][:3]).max())), max(track, float(np.abs(pos - g["des_pos"][k]).max()))
    ref_track = float(np.abs(g["state_next"][:, :3] - g["des_pos"]).max())         # the reference's own worst tracking error on this run
    assert err < 0.5 and track < 1.25 * ref_track, (err, track, ref_track)
    assert np.abs(pos - g["waypoints"][-1]).max() < 5e-2                           # it ends hovering at the last waypoint


@pytest.mark.parametrize("vehicle", ["quad", "hexa"])
def test_pid_policy_kernel_equals_the_torch_restatement(vehicle):
    """amenv_pid_policy (one launch) against PidWaypointPolicy's torch path, both in fp64 on the same observation stream from the HIP
    env (driven by the kernel's actions), episode starts included."""
    n = 512
    env = amd.GpuWaypointEnv(n, vehicle=vehicle, seed=4)
    hip = PidWaypointPolicy.for_env(env, dtype=torch.float64)
    assert hip.pstate is not None
    v = env.cfg.vehicle
    ref = PidWaypointPolicy(n, dt=env.cfg.task.dt, mass=v.mass, g=v.g, moment_scale=v.moment_scale,
                            inertia_diag=(v.inertia[0], v.inertia[4], v.inertia[8]), speed=0.6, dtype=torch.float64)
    obs, done, worst, ends = env.reset(), None, 0.0, 0
    for _ in range(700):
        a = hip.predict(obs, done)
        b = ref.predict(obs.cpu(), None if done is None else done.cpu())
        worst = max(worst, float((a.cpu() - b).abs().max()))
        obs, _, done, _ = env.step(a)
        ends += int(done.sum())
    assert worst < 5e-6, worst            # fp64 arithmetic on both sides, fp32 action rows
    assert ends > 50                      # episode ends (new trajectories, integrals cleared) were part of the stream
    assert torch.allclose(hip.pstate[:, 7:13].cpu(), ref.pid.integral, rtol=1e-9, atol=1e-12)
    assert torch.allclose(hip.pstate[:, 0].cpu(), ref.t, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("vehicle,floor", [("quad", 0.8), ("hexa", 0.75), ("hexa_arm", 0.8)])
def test_pid_baseline_flies_the_gpu_env(vehicle, floor):
    """Closed loop on the HIP env, everything resident on the GPU, two launches per control step (amenv_pid_policy, amenv_step).  With the
    arm the position loop tracks the TOOL point (the task's point: forward kinematics in obs[26:29]) and the joints are held at home:
    an action source independent of any trained policy that brings the tool point of the SDF-derived hexacopter + arm to the waypoint."""
    env = amd.GpuWaypointEnv(1024, vehicle=vehicle, seed=3)
    pol = PidWaypointPolicy.for_env(env)
    assert pol.pstate is not None and pol.pstate.dtype == torch.float32       # the fp32 HIP kernel is what flies
    assert pol.tool_mode == (vehicle == "hexa_arm") and pol.act_dim == env.act_dim
    obs = env.reset()
    env.stats(reset=True)
    done = None
    for _ in range(1700):
        obs, _, done, _ = env.step(pol.predict(obs, done))
    s = env.stats()
    assert s["episodes"] > 1000 and s["success"] / s["episodes"] > floor
    assert s["nonfinite"] == 0


def test_pid_policy_is_one_launch_on_device():
    env = amd.GpuWaypointEnv(256, seed=1)
    pol = PidWaypointPolicy.for_env(env)
    a = pol.predict(env.reset())
    assert a.is_cuda and a.dtype == torch.float32 and a.shape == (256, 4)
    assert bool((a[:, 0] >= 0).all()) and bool((a[:, 0] <= 2).all()) and bool((a[:, 1:].abs() <= 1).all())
    # the first action of an episode at rest on the trajectory start is hover thrust, no moments
    assert torch.allclose(a[:, 0], torch.ones(256, device=a.device), atol=1e-4) and float(a[:, 1:].abs().max()) < 1e-3
    # graph-capturable: no allocation-free requirement on the host wrapper, but the launch itself only enqueues
    assert float(pol.pstate[:, 13].max()) == 0.0 and float(pol.pstate[:, 0].min()) > 0.0
    # bad arguments are refused, not launched
    import ctypes as C
    p = amd._lib.PidPolicyParams(pid=pol.pid.params(), speed=0.6, moment_scale=0.1, obs_dim=19, act_dim=4)
    assert amd._lib.load().amenv_pid_policy(C.byref(p), 0, C.c_void_p(a.data_ptr()), None, C.c_void_p(pol.pstate.data_ptr()), C.c_void_p(a.data_ptr()), 256, None) == -1


def test_runsim_example_runs():
    """examples/runsim_gpu.py (the reference's PID demo loop for N vehicles on the GPU) end to end: every vehicle flies its own helix within
    tracking distance of its trajectory and the ones that finish hover on their last waypoint."""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "runsim_gpu.py"), "--vehicles", "256", "--steps", "1500"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    m = re.search(r"worst tracking error per vehicle: median ([0-9.]+) m, max ([0-9.]+) m", r.stdout)
    assert m and float(m.group(1)) < 0.5 and float(m.group(2)) < 1.5, r.stdout
    m2 = re.search(r"(\d+) vehicles finished their trajectory: distance to the last waypoint median ([0-9.]+) m", r.stdout)
    assert m2 and int(m2.group(1)) > 10 and float(m2.group(2)) < 0.05, r.stdout
