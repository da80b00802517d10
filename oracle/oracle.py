"""ctypes binding of the CPU oracle (oracle/amenv_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libamenv_oracle.so")

MAX_ROTORS, MAX_WAYPOINTS, MAX_JOINTS = 8, 4, 3

# info bits / flags (include/amenv.h)
INFO_TERMINATED, INFO_TRUNCATED, INFO_SUCCESS, INFO_STOPPED, INFO_CRASHED, INFO_OOB, INFO_NONFINITE, INFO_WAS_RESET = (1 << i for i in range(8))
FLAG_AUTO_RESET, FLAG_NAN_GUARD = 1, 2
F32, F64 = 0, 1
F_FINAL_YAW, F_LAST_DISTANCE, F_EP_RETURN, F_WP0 = 13, 14, 15, 16
I_STEP, I_COUNTER, I_FLAGS, I_EPISODE, I_NFIELDS = 0, 1, 2, 3, 4
FLAGBIT_FWR, FLAGBIT_COUNTER_ACTIVE = 256, 512
TASK_V2_SCALED20, TASK_V1_SCALED17, TASK_V1_RAW17 = 0, 1, 2
EE_TASK_BASE, EE_TASK_TOOL = 0, 1


class Vehicle(C.Structure):
    _fields_ = [
        ("n_rotors", C.c_int32), ("n_joints", C.c_int32), ("mass", C.c_double), ("g", C.c_double),
        ("inertia", C.c_double * 9), ("inv_inertia", C.c_double * 9),
        ("alloc", C.c_double * (MAX_ROTORS * 4)), ("mix", C.c_double * (4 * MAX_ROTORS)),
        ("t_min", C.c_double * MAX_ROTORS), ("t_max", C.c_double * MAX_ROTORS), ("moment_scale", C.c_double),
        ("joint_origin", C.c_double * (MAX_JOINTS * 3)), ("joint_axis", C.c_double * (MAX_JOINTS * 3)),
        ("link_mass", C.c_double * MAX_JOINTS), ("link_com", C.c_double * (MAX_JOINTS * 3)),
        ("link_inertia", C.c_double * (MAX_JOINTS * 9)),
        ("joint_kp", C.c_double), ("joint_kd", C.c_double), ("joint_acc_max", C.c_double), ("joint_reserved", C.c_double),
        ("joint_limit", C.c_double * (MAX_JOINTS * 2)), ("tool_offset", C.c_double * 3),
    ]


class Task(C.Structure):
    _fields_ = [
        ("variant", C.c_int32), ("num_waypoints", C.c_int32), ("max_episode_steps", C.c_int32),
        ("counter_limit", C.c_int32), ("rk4_substeps", C.c_int32), ("ee_task", C.c_int32), ("dt", C.c_double),
        ("traj_sin", C.c_double * MAX_WAYPOINTS), ("traj_cos", C.c_double * MAX_WAYPOINTS),
    ]


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32), ("num_envs", C.c_int32), ("dtype", C.c_int32),
        ("flags", C.c_uint32), ("block_size", C.c_int32), ("step_kernel", C.c_int32), ("reserved1", C.c_int32), ("seed", C.c_uint64), ("env_id_offset", C.c_int64),
        ("vehicle", Vehicle), ("task", Task),
    ]


def build(force=False):
    """Compile the oracle with gcc (seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "amenv_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "amenv.h")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libamenv_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        P = C.c_void_p
        L.orc_reference_quad.argtypes = [C.POINTER(Config), C.c_int32]
        L.orc_set_num_waypoints.argtypes = [C.POINTER(Config), C.c_int32]
        L.orc_n_float_fields.argtypes = [C.POINTER(Config)]
        L.orc_obs_dim.argtypes = [C.POINTER(Config)]
        L.orc_act_dim.argtypes = [C.POINTER(Config)]
        L.orc_arm_rhs.argtypes = [C.POINTER(Config), P, C.c_double, P, P, P]
        L.orc_arm_dynamics_step.argtypes = [C.POINTER(Config), P, P, P]
        L.orc_dynamics_step.argtypes = [C.POINTER(Config), P, P, P]
        L.orc_reset.argtypes = [C.POINTER(Config), P, P, P, P]
        L.orc_observe.argtypes = [C.POINTER(Config), P, P, P]
        L.orc_step.argtypes = [C.POINTER(Config)] + [P] * 10 + [C.c_int]
        L.orc_rollout.argtypes = [C.POINTER(Config), P, P, C.c_int, P, P, C.c_int]
        L.orc_ee_position.argtypes = [C.POINTER(Config), P, P]
        L.orc_ee_positions.argtypes = [C.POINTER(Config), P, P]
        L.orc_philox.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, P]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def reference_quad_config(num_envs=1, seed=0, flags=FLAG_AUTO_RESET, num_waypoints=1, variant=TASK_V2_SCALED20):
    cfg = Config()
    lib().orc_reference_quad(C.byref(cfg), num_envs)
    cfg.seed = seed
    cfg.flags = flags
    if variant != TASK_V2_SCALED20:        # v1 envs: 17-D obs, 1..2 waypoints per episode, 1200-step limit
        cfg.task.variant = variant
        cfg.task.num_waypoints = 2         # v1/rl_env_scaledObs.py:38 (storage bound; per-episode K is drawn at reset)
        cfg.task.max_episode_steps = 1200  # v1/rl_env_scaledObs.py:43
    elif num_waypoints != 1:
        lib().orc_set_num_waypoints(C.byref(cfg), num_waypoints)
    return cfg


class OracleEnv:
    """Batched CPU oracle with the same SoA state blob as libamenv (fstate fp64 [NF,N], istate i32 [4,N])."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.n = cfg.num_envs
        self.nf = lib().orc_n_float_fields(C.byref(cfg))
        self.fstate = np.zeros((self.nf, self.n), np.float64)
        self.istate = np.zeros((I_NFIELDS, self.n), np.int32)
        self.obs_dim, self.act_dim = lib().orc_obs_dim(C.byref(cfg)), lib().orc_act_dim(C.byref(cfg))

    def reset(self, mask=None):
        obs = np.zeros((self.n, self.obs_dim), np.float32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        lib().orc_reset(C.byref(self.cfg), _p(self.fstate), _p(self.istate), _p(m), _p(obs))
        return obs

    def observe(self):
        obs = np.zeros((self.n, self.obs_dim), np.float32)
        lib().orc_observe(C.byref(self.cfg), _p(self.fstate), _p(self.istate), _p(obs))
        return obs

    def ee_position(self):
        """World position of the arm's tool point for every env, [N,3] fp64 (forward kinematics; body origin without an arm)."""
        out = np.zeros((self.n, 3), np.float64)
        lib().orc_ee_positions(C.byref(self.cfg), _p(self.fstate), _p(out))
        return out

    def step(self, actions, nthreads=1):
        a = np.ascontiguousarray(actions, np.float32).reshape(self.n, self.act_dim)
        out = dict(
            obs=np.zeros((self.n, self.obs_dim), np.float32), reward=np.zeros(self.n, np.float64),
            done=np.zeros(self.n, np.uint8), info=np.zeros(self.n, np.uint32),
            terminal_obs=np.full((self.n, self.obs_dim), np.nan, np.float32),
            ep_return=np.full(self.n, np.nan, np.float32), ep_len=np.full(self.n, -1, np.int32),
        )
        lib().orc_step(C.byref(self.cfg), _p(self.fstate), _p(self.istate), _p(a), _p(out["obs"]), _p(out["reward"]),
                       _p(out["done"]), _p(out["info"]), _p(out["terminal_obs"]), _p(out["ep_return"]), _p(out["ep_len"]), nthreads)
        return out

    def rollout(self, actions, nthreads=1):
        a = np.ascontiguousarray(actions, np.float32)
        T = a.shape[0]
        assert a.shape == (T, self.n, self.act_dim)
        rs = np.zeros(self.n, np.float64)
        lib().orc_rollout(C.byref(self.cfg), _p(self.fstate), _p(self.istate), T, _p(a), _p(rs), nthreads)
        return rs

    # ---- helpers to inject a golden-vector row as the state of env i -------------------------
    def set_env(self, i, state13, waypoints, final_yaw, last_distance, waypoint_index, fwr, counter, counter_activated,
                current_step, ep_return=0.0, episode=1, k_env=0):
        self.fstate[0:13, i] = state13
        self.fstate[F_FINAL_YAW, i] = final_yaw
        self.fstate[F_LAST_DISTANCE, i] = -1.0 if (last_distance is None or np.isnan(last_distance)) else last_distance
        self.fstate[F_EP_RETURN, i] = ep_return
        wp = np.asarray(waypoints, np.float64).reshape(-1, 3)
        self.fstate[F_WP0:F_WP0 + 3 * wp.shape[0], i] = wp.reshape(-1)
        self.istate[I_STEP, i] = current_step
        self.istate[I_COUNTER, i] = counter
        self.istate[I_FLAGS, i] = (int(waypoint_index) & 15) | ((int(k_env) & 15) << 4) | (FLAGBIT_FWR if fwr else 0) | (FLAGBIT_COUNTER_ACTIVE if counter_activated else 0)
        self.istate[I_EPISODE, i] = episode


def dynamics_step(cfg, state13, action):
    s = np.array(state13, np.float64).copy()
    a = np.ascontiguousarray(action, np.float32)
    wrench = np.zeros(8, np.float64)
    lib().orc_dynamics_step(C.byref(cfg), _p(s), _p(a), _p(wrench))
    return s, wrench


def arm_rhs(cfg, s19, F, M, th_cmd):
    s = np.ascontiguousarray(s19, np.float64); M = np.ascontiguousarray(M, np.float64); c = np.ascontiguousarray(th_cmd, np.float64)
    d = np.zeros(19)
    lib().orc_arm_rhs(C.byref(cfg), _p(s), float(F), _p(M), _p(c), _p(d))
    return d


def arm_dynamics_step(cfg, s19, action7):
    s = np.array(s19, np.float64).copy(); a = np.ascontiguousarray(action7, np.float32); w = np.zeros(8)
    lib().orc_arm_dynamics_step(C.byref(cfg), _p(s), _p(a), _p(w))
    return s, w


def ee_position(cfg, s19):
    s = np.ascontiguousarray(s19, np.float64); out = np.zeros(3)
    lib().orc_ee_position(C.byref(cfg), _p(s), _p(out))
    return out


def philox(seed, gid, episode, block):
    out = np.zeros(4, np.uint32)
    lib().orc_philox(seed, gid, episode, block, _p(out))
    return out


def max_threads():
    return lib().orc_max_threads()


def gae_reference(rewards, values, dones, last_values, gamma, gae_lambda):
    """numpy restatement of SB3 2.6.0 `RolloutBuffer.compute_returns_and_advantage` (third-party; the reference reaches
    it through `PPO.learn`, v2/rl_train.py:56) in float64 on time-major [T, N] arrays.  `dones[t]` = episode ended at step
    t, i.e. SB3's `episode_starts[t + 1]` and, for the last row, its `dones` argument."""
    r, v = np.asarray(rewards, np.float64), np.asarray(values, np.float64)
    nnt = 1.0 - (np.asarray(dones) != 0).astype(np.float64)
    T = r.shape[0]
    adv = np.zeros_like(r)
    last, next_v = np.zeros(r.shape[1]), np.asarray(last_values, np.float64)
    for t in reversed(range(T)):
        delta = r[t] + gamma * next_v * nnt[t] - v[t]
        last = delta + gamma * gae_lambda * nnt[t] * last
        adv[t] = last
        next_v = v[t]
    return adv, adv + v
