"""PID + minimum-snap baseline controller, batched.

For N vehicles at once.  On CUDA tensors every operation below is ONE launch of a hand-written HIP kernel through the C ABI
(`amenv_pid_run`, `amenv_minsnap_solve` / `_eval`, `amenv_pid_policy`; csrc/amenv_baseline.hpp) -- no library is missing silently: without
libamenv.so the call raises.  On CPU tensors the same arithmetic runs as the torch restatement in this file, which is what the CPU
tests pin to the reference's recorded run and what the GPU tests compare the kernels with.
  * `PidController`     -- `PID Controller/pid_controller.py:37-115` (gains :16-21, integral clamp :34,66-67,107-108)
  * `MinSnapTrajectory` -- `PID Controller/trajGen3D.py`: `MST` (:211-292, 7th-order segments, the constraint rows in the
                           same order), `get_poly_cc` (:189-209), `generate_trajectory` (:76-187; its yaw bookkeeping ends
                           in `yaw = 0; yawdot = 0` at :183-184, which is what is returned here)
  * `PidWaypointPolicy` -- the two wired to the waypoint environment as an independent closed-loop action source: each
                           episode flies a one-segment minimum-snap trajectory from the start position to the waypoint, the
                           way `PID Controller/runsim.py:26-31` flies its waypoint list.
Pinned by tests/golden/pid_helix.npz (tools/gen_golden_pid.py: the unmodified reference modules run as runsim.py runs them).
"""

import ctypes as C

import torch

from . import _lib

# pid_controller.py:16-21
GAINS = dict(x=(3.0, 30.0, 1.0), y=(3.0, 30.0, 1.0), z=(1000.0, 200.0, 10.0),
             phi=(160.0, 3.0, 1.0), theta=(160.0, 3.0, 1.0), psi=(80.0, 5.0, 1.0))   # (k_p, k_d, k_i)
MAX_INTEGRAL = 100.0                                                                   # pid_controller.py:34


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _dtype_code(t):
    return _lib.F64 if t.dtype == torch.float64 else _lib.F32


def _check(rc, what):
    if rc != 0:
        raise _lib.AmenvError(f"{what} failed ({rc})")


def rot_to_rpy(q):
    """`Quadcopter.attitude()` (model/quadcopter.py:57-59): `RotToRPY(Quaternion(q).as_rotation_matrix())`
    (utils/utils.py:11-15) from the entries of R(q / |q|) it uses.  q: [N, 4] (w, x, y, z) -> phi, theta, psi [N]."""
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    r12 = 2 * (y * z - w * x)
    r02 = 2 * (x * z + w * y)
    r22 = 1 - 2 * (x * x + y * y)
    r10 = 2 * (x * y + w * z)
    r11 = 1 - 2 * (x * x + z * z)
    phi = torch.asin(r12.clamp(-1.0, 1.0))
    c = torch.cos(phi)
    return phi, torch.atan2(-r02 / c, r22 / c), torch.atan2(-r10 / c, r11 / c)


class PidController:
    """pid_controller.run for N vehicles: cascaded PID, position loop -> commanded acceleration -> thrust and desired
    roll / pitch -> attitude loop -> moments.  Integral memory [N, 6] lives on the tensors' device."""

    def __init__(self, num_envs, dt, mass=0.18, g=9.81, device="cpu", dtype=torch.float64, gains=None):
        self.n, self.dt, self.mass, self.g = int(num_envs), float(dt), float(mass), float(g)
        self.gains = dict(GAINS if gains is None else gains)
        self.integral = torch.zeros(self.n, 6, device=device, dtype=dtype)   # x y z phi theta psi

    def params(self):
        """-> the C ABI's amenv_pid_params for this controller."""
        p = _lib.PidParams(dt=self.dt, mass=self.mass, g=self.g, max_integral=MAX_INTEGRAL)
        for k, name in enumerate(("x", "y", "z", "phi", "theta", "psi")):
            p.gain[3 * k:3 * k + 3] = self.gains[name]
        return p

    def run_state(self, state, des):
        """state [N, 13] (`Quadcopter.state`: p, v, q, body rates), des [N, 11] (pos, vel, acc, yaw, yaw rate) -> F [N], M [N, 3],
        rpy [N, 3].  CUDA tensors: one launch of `amenv_pid_run` (fp64 or fp32 after the integral memory's dtype)."""
        I = self.integral
        state, des = state.to(I.dtype).contiguous(), des.to(I.dtype).contiguous()
        if not I.is_cuda:
            rpy = rot_to_rpy(state[:, 6:10])
            F, M = self.run(state[:, 0:3], state[:, 3:6], rpy, state[:, 10:13], des[:, 0:3], des[:, 3:6], des[:, 6:9], des[:, 9], des[:, 10])
            return F, M, torch.stack(rpy, -1)
        n = state.shape[0]
        F, M, rpy = I.new_empty(n), I.new_empty(n, 3), I.new_empty(n, 3)
        _check(_lib.load().amenv_pid_run(C.byref(self.params()), _dtype_code(I), _p(state), _p(des), _p(I), _p(F), _p(M), _p(rpy), n, _stream(I)),
               "amenv_pid_run")
        return F, M, rpy

    def reset(self, mask=None):
        if mask is None:
            self.integral.zero_()
        else:
            self.integral.mul_((~mask.bool()).to(self.integral.dtype).unsqueeze(-1))

    def run(self, pos, vel, rpy, omega, des_pos, des_vel, des_acc, des_yaw, des_yawdot):
        """-> F [N] (N), M [N, 3] (N m), exactly the reference's arithmetic (:51-113)."""
        G, I, dt = self.gains, self.integral, self.dt
        err_p = des_pos - pos                                            # :51-53
        err_v = des_vel - vel                                            # :54-56
        I[:, 0:3] = (I[:, 0:3] + err_p * dt).clamp(-MAX_INTEGRAL, MAX_INTEGRAL)   # :59-67
        acc = []
        for k, name in enumerate("xyz"):                                 # :70-83
            kp, kd, ki = G[name]
            acc.append(des_acc[:, k] + kd * err_v[:, k] + kp * err_p[:, k] + ki * I[:, k])
        F = self.mass * (self.g + acc[2])                                # :86
        s, c = torch.sin(des_yaw), torch.cos(des_yaw)
        des_phi = 1.0 / self.g * (acc[0] * s - acc[1] * c)               # :89
        des_theta = 1.0 / self.g * (acc[0] * c + acc[1] * s)             # :90
        phi, theta, psi = rpy
        err_a = torch.stack([des_phi - phi, des_theta - theta, des_yaw - psi], -1)        # :91-96
        err_w = torch.stack([-omega[:, 0], -omega[:, 1], des_yawdot - omega[:, 2]], -1)   # :97-99
        I[:, 3:6] = (I[:, 3:6] + err_a * dt).clamp(-MAX_INTEGRAL, MAX_INTEGRAL)           # :102-108
        M = []
        for k, name in enumerate(("phi", "theta", "psi")):               # :111-115
            kp, kd, ki = G[name]
            M.append(kp * err_a[:, k] + kd * err_w[:, k] + ki * I[:, 3 + k])
        return F, torch.stack(M, -1)


def poly_cc(n, k, t, dtype=torch.float64, device="cpu"):
    """trajGen3D.get_poly_cc: coefficients of the k-th derivative of sum_i a_i t^i (i < n) at t.  t: tensor [...] -> [..., n]."""
    t = torch.as_tensor(t, dtype=dtype, device=device)
    i = torch.arange(n, dtype=dtype, device=device)
    coef = torch.ones(n, dtype=dtype, device=device)
    for j in range(k):
        coef = coef * (i - j).clamp(min=0.0)
    power = (i - k).clamp(min=0.0)
    return coef * t.unsqueeze(-1) ** power


class MinSnapTrajectory:
    """Minimum-snap polynomial trajectory through waypoints [..., n+1, 3] (batched over leading dims), trajGen3D.MST."""

    def __init__(self, waypoints, speed=1.2):
        w = torch.as_tensor(waypoints, dtype=torch.float64)
        self.batched = w.dim() == 3
        if not self.batched:
            w = w.unsqueeze(0)
        w = w.contiguous()
        self.waypoints, self.speed = w, float(speed)
        n = w.shape[1] - 1
        self.n = n
        dev = w.device
        if w.is_cuda:     # amenv_minsnap_solve: constraint matrix inverted on the device, one matrix product per trajectory
            lib, nb = _lib.load(), w.shape[0]
            self.coeff, self.T, self.S = w.new_empty(nb, 8 * n, 3), w.new_empty(nb, n), w.new_empty(nb, n + 1)
            ws = torch.empty(lib.amenv_minsnap_workspace_bytes(n) // 8, dtype=torch.float64, device=dev)
            _check(lib.amenv_minsnap_solve(n, nb, self.speed, _p(w), _p(self.coeff), _p(self.T), _p(self.S), _p(ws), _stream(w)), "amenv_minsnap_solve")
            return
        A = torch.zeros(8 * n, 8 * n, dtype=torch.float64, device=dev)
        c0 = [poly_cc(8, k, 0.0, device=dev) for k in range(7)]
        c1 = [poly_cc(8, k, 1.0, device=dev) for k in range(7)]
        for i in range(n):                                    # constraints 1, 2: segment end points (:262-268)
            A[i, 8 * i:8 * i + 8] = c0[0]
            A[i + n, 8 * i:8 * i + 8] = c1[0]
        for k in range(1, 4):                                 # 3, 4: rest at both ends (:270-276)
            A[2 * n + k - 1, :8] = c0[k]
            A[2 * n + 3 + k - 1, -8:] = c1[k]
        for i in range(n - 1):                                # 5: derivatives 1..6 continuous at the knots (:278-281)
            for k in range(1, 7):
                A[2 * n + 6 + i * 6 + k - 1, 8 * i:8 * i + 16] = torch.cat([c1[k], -c0[k]])
        B = torch.zeros(w.shape[0], 8 * n, 3, dtype=torch.float64, device=dev)
        B[:, :n] = w[:, :-1]
        B[:, n:2 * n] = w[:, 1:]
        self.coeff = torch.linalg.solve(A, B)                 # [batch, 8n, 3]   (x, y, z share A)
        self.T = (w[:, :-1] - w[:, 1:]).norm(dim=-1) / self.speed          # segment times (:97-99)
        self.S = torch.cat([torch.zeros_like(self.T[:, :1]), self.T.cumsum(-1)], -1)

    def desired_state(self, t):
        """generate_trajectory(t, v, ...) -> pos, vel, acc [batch, 3], yaw, yawdot [batch] (both 0, :183-184)."""
        w, S, T = self.waypoints, self.S, self.T
        t = torch.as_tensor(t, dtype=torch.float64, device=w.device).expand(w.shape[0])
        if w.is_cuda:
            des = self.evaluate(t.contiguous())
            out = (des[:, 0:3], des[:, 3:6], des[:, 6:9], des[:, 9], des[:, 10])
            return out if self.batched else tuple(o[0] for o in out)
        idx = ((t.unsqueeze(-1) >= S).sum(-1) - 1).clamp(0, self.n - 1)             # :105
        Ti = T.gather(1, idx.unsqueeze(-1)).squeeze(-1)
        scale = (t - S.gather(1, idx.unsqueeze(-1)).squeeze(-1)) / Ti                 # :123
        seg = self.coeff.gather(1, (8 * idx).view(-1, 1, 1) + torch.arange(8, device=w.device).view(1, 8, 1).expand(w.shape[0], 8, 3))
        pos = torch.einsum("bk,bkc->bc", poly_cc(8, 0, scale, device=w.device), seg)
        vel = torch.einsum("bk,bkc->bc", poly_cc(8, 1, scale, device=w.device), seg) / Ti.unsqueeze(-1)
        acc = torch.einsum("bk,bkc->bc", poly_cc(8, 2, scale, device=w.device), seg) / (Ti * Ti).unsqueeze(-1)
        first, after = (t == 0).unsqueeze(-1), (t > S[:, -1]).unsqueeze(-1)           # :108,121
        zero = torch.zeros_like(pos)
        pos = torch.where(first, w[:, 0], torch.where(after, w[:, -1], pos))
        vel = torch.where(first | after, zero, vel)
        acc = torch.where(first | after, zero, acc)
        yaw = torch.zeros_like(t)
        out = (pos, vel, acc, yaw, yaw.clone())
        return out if self.batched else tuple(o[0] for o in out)


    def _evaluate(self, t, traj=None, dtype=torch.float64):
        """CUDA only: `amenv_minsnap_eval` for m (trajectory, time) queries -> des [m, 11] (pos, vel, acc, yaw, yaw rate) of `dtype`."""
        w = self.waypoints
        t = torch.as_tensor(t, dtype=torch.float64, device=w.device).contiguous()
        if traj is not None:
            traj = torch.as_tensor(traj, dtype=torch.int64, device=w.device).contiguous()
        des = torch.empty(t.numel(), 11, dtype=dtype, device=w.device)
        _check(_lib.load().amenv_minsnap_eval(self.n, t.numel(), _p(self.coeff), _p(self.T), _p(self.S), _p(w), _p(traj), _p(t), _dtype_code(des), _p(des),
                                              _stream(w)), "amenv_minsnap_eval")
        return des

    def evaluate(self, t, traj=None, dtype=torch.float64):
        """m queries at once: des [m, 11]; query i = trajectory traj[i] (None: trajectory i) at time t[i]."""
        if self.waypoints.is_cuda:
            return self._evaluate(t, traj, dtype)
        t = torch.as_tensor(t, dtype=torch.float64)
        idx = torch.arange(t.numel()) if traj is None else torch.as_tensor(traj, dtype=torch.int64)
        one = MinSnapTrajectory.__new__(MinSnapTrajectory)
        one.__dict__.update(self.__dict__)
        one.batched = True
        one.waypoints, one.coeff, one.T, one.S = self.waypoints[idx], self.coeff[idx], self.T[idx], self.S[idx]
        p, v, a, yaw, yawdot = one.desired_state(t)
        return torch.cat([p, v, a, yaw.unsqueeze(-1), yawdot.unsqueeze(-1)], -1).to(dtype)


class PidWaypointPolicy:
    """Closed-loop baseline for the waypoint environment: `predict(obs, done)` -> actions [N, 4] from the 20-D observation
    alone (position, velocity, quaternion, body rates and the vector to the current waypoint are all in it,
    v2/rl_env_scaledObs.py:98-121).  Per episode: a one-segment minimum-snap trajectory (rest to rest, 7th order; for one
    segment MST's solution is s(tau) = 35 tau^4 - 84 tau^5 + 70 tau^6 - 20 tau^7) from where the episode started to the
    waypoint at `speed` m/s (0.6 by default: the reference's 1.2 m/s is flown on mostly vertical helix
    segments; on this task's lateral segments the committed gains hold up to ~0.8 m/s), tracked by `PidController`; moments are scaled by the vehicle's inertia relative to the
    reference quadrotor's so the same gains fly the hexacopter."""

    QUAD_INERTIA = (2.5e-4, 2.32e-4, 3.738e-4)     # simul_files/model/params.py

    def __init__(self, num_envs, dt=1.0 / 200.0, mass=0.18, g=9.81, moment_scale=0.1, inertia_diag=None, speed=0.6,
                 device="cpu", dtype=torch.float32, gains=None, act_dim=4, tool_mode=False):
        self.pid = PidController(num_envs, dt, mass, g, device, dtype, gains)
        self.act_dim, self.tool_mode = int(act_dim), bool(tool_mode)
        # CUDA: the whole predict() is one launch of amenv_pid_policy on this state block [N, 14] = t, start, goal, integrals, fresh
        self.pstate = None
        if torch.device(device).type == "cuda":
            self.pstate = torch.zeros(num_envs, 14, device=device, dtype=dtype)
            self.pstate[:, 13] = 1.0
            self.inertia_ratio = [a / b for a, b in zip(self.QUAD_INERTIA if inertia_diag is None else inertia_diag, self.QUAD_INERTIA)]
        self.dt, self.speed, self.mass, self.g, self.moment_scale = float(dt), float(speed), float(mass), float(g), float(moment_scale)
        inertia = self.QUAD_INERTIA if inertia_diag is None else inertia_diag
        self.m_gain = torch.tensor([a / b for a, b in zip(inertia, self.QUAD_INERTIA)], device=device, dtype=dtype)
        z = lambda *s: torch.zeros(*s, device=device, dtype=dtype)  # noqa: E731
        self.t, self.start, self.goal = z(num_envs), z(num_envs, 3), z(num_envs, 3)
        self.fresh = torch.ones(num_envs, dtype=torch.bool, device=device)
        self.low = torch.tensor([0.0, -1.0, -1.0, -1.0], device=device, dtype=dtype)
        self.high = torch.tensor([2.0, 1.0, 1.0, 1.0], device=device, dtype=dtype)

    @classmethod
    def for_env(cls, env, speed=0.6, dtype=torch.float32):
        v = env.cfg.vehicle
        arm = v.n_joints > 0
        return cls(env.num_envs, dt=env.cfg.task.dt, mass=v.mass, g=v.g, moment_scale=v.moment_scale,
                   inertia_diag=(v.inertia[0], v.inertia[4], v.inertia[8]), speed=speed, device=env.device, dtype=dtype,
                   act_dim=env.act_dim, tool_mode=arm and env.cfg.task.ee_task == _lib.EE_TASK_TOOL)

    @torch.no_grad()
    def predict(self, obs, done=None):
        """obs [N, >=16] f32 (v2 layout), done [N] from the previous step (those envs were auto-reset: new episode)."""
        if self.pstate is not None:
            return self._predict_hip(obs, done)
        if self.act_dim != 4 or self.tool_mode:
            raise NotImplementedError("the torch restatement covers the rigid vehicles (4 actions); the arm runs on the HIP kernel")
        o = obs.to(self.t.dtype)
        if done is not None:
            self.fresh |= done.bool()
        pos, vel, quat, omega = o[:, 0:3] * 10.0, o[:, 3:6] * 5.0, o[:, 6:10], o[:, 10:13] * 5.0
        goal = pos + o[:, 13:16] * 2.0
        f = self.fresh                                   # masked updates, unconditional: no host sync on the GPU
        fm = f.unsqueeze(-1)
        self.start = torch.where(fm, pos, self.start)
        self.goal = torch.where(fm, goal, self.goal)
        self.t = torch.where(f, torch.zeros_like(self.t), self.t)
        self.pid.reset(f)
        self.fresh = torch.zeros_like(f)
        # waypoint switched inside an episode (multi-waypoint tasks): start a new segment from the current position
        moved = (goal - self.goal).norm(dim=-1) > 1e-3
        mm = moved.unsqueeze(-1)
        self.start = torch.where(mm, pos, self.start)
        self.goal = torch.where(mm, goal, self.goal)
        self.t = torch.where(moved, torch.zeros_like(self.t), self.t)
        d = self.goal - self.start
        T = (d.norm(dim=-1) / self.speed).clamp(min=self.dt)
        tau = (self.t / T).clamp(0.0, 1.0)
        t2 = tau * tau
        t3, t4 = t2 * tau, t2 * t2
        s0 = t4 * (35.0 + tau * (-84.0 + tau * (70.0 - 20.0 * tau)))
        s1 = t3 * (140.0 + tau * (-420.0 + tau * (420.0 - 140.0 * tau))) / T
        s2 = t2 * (420.0 + tau * (-1680.0 + tau * (2100.0 - 840.0 * tau))) / (T * T)
        des_pos = self.start + d * s0.unsqueeze(-1)
        des_vel, des_acc = d * s1.unsqueeze(-1), d * s2.unsqueeze(-1)
        zero = torch.zeros_like(self.t)
        F, M = self.pid.run(pos, vel, rot_to_rpy(quat), omega, des_pos, des_vel, des_acc, zero, zero)
        self.t = self.t + self.dt
        # The reference hands F, M to the mixer unclipped (runsim.py:30) and lets the per-rotor clamp sort it out; the env's
        # action box clips each moment separately, which would let a saturated yaw demand (the largest per-rotor share)
        # drown roll / pitch.  Scale the moment VECTOR into the box instead: direction and ratios are kept.
        am = M * self.m_gain / self.moment_scale
        am = am / am.abs().amax(dim=-1, keepdim=True).clamp(min=1.0)
        a = torch.cat([(F / (self.mass * self.g)).unsqueeze(-1), am], -1)
        return torch.minimum(torch.maximum(a, self.low), self.high).to(torch.float32)

    def _predict_hip(self, obs, done):
        n, pid = self.pstate.shape[0], self.pid
        obs = obs.contiguous()
        assert obs.is_cuda and obs.dtype == torch.float32 and obs.shape[0] == n and obs.shape[1] >= (29 if self.tool_mode else 20)
        if done is not None:
            done = done.to(torch.uint8).contiguous()
        p = _lib.PidPolicyParams(pid=pid.params(), speed=self.speed, moment_scale=self.moment_scale, obs_dim=obs.shape[1], act_dim=self.act_dim,
                                 tool_mode=int(self.tool_mode))
        p.inertia_ratio[:] = self.inertia_ratio
        act = torch.empty(n, self.act_dim, device=obs.device, dtype=torch.float32)
        _check(_lib.load().amenv_pid_policy(C.byref(p), _dtype_code(self.pstate), _p(obs), _p(done), _p(self.pstate), _p(act), n, _stream(obs)),
               "amenv_pid_policy")
        return act
