"""GpuWaypointEnv -- the torch-native batched environment over libamenv.so.

N independent copies of the reference's `WaypointQuadEnv` (v2/rl_env_scaledObs.py:9-231)
live in the HBM of one MI355X; `step()` is ONE HIP kernel launch on torch's current stream.
PyTorch is only the device container here (buffers, streams): all arithmetic is in the HIP
kernels behind the C ABI (include/amenv.h).  There is no CPU path.
"""
import ctypes as C

import torch

from . import _lib as L


def _dev_index(device):
    if isinstance(device, int):
        return device
    d = torch.device(device)
    if d.type != "cuda":
        raise L.AmenvError(f"GpuWaypointEnv needs a GPU device (got {d}); there is no CPU fallback")
    return d.index if d.index is not None else torch.cuda.current_device()


class GpuWaypointEnv:
    """Batched waypoint environment resident on one GPU.

    step(actions[N,4] f32 cuda) -> (obs[N,20] f32, reward[N], done[N] u8, info_bits[N] i32)
    Returned tensors are persistent output buffers, overwritten by the next call (zero-copy).
    After a step, for envs with done != 0: `terminal_obs[i]`, `ep_return[i]`, `ep_len[i]` hold
    the SB3 `terminal_observation` and Monitor `episode` r / l values.
    """

    def __init__(self, num_envs, device=0, vehicle="quad", seed=0, dtype="f32", auto_reset=True, nan_guard=False,
                 num_waypoints=1, env_id_offset=0, block_size=0, max_episode_steps=None, counter_limit=None,
                 rk4_substeps=1, task="v2", config=None, kernel="auto", ee_task=None):
        self.lib = L.load()
        self.device_index = _dev_index(device)
        self.device = torch.device("cuda", self.device_index)
        if config is None:
            cfg = L.default_config(vehicle, num_envs, task)   # task: "v2" | "v1_scaled" | "v1_raw" (which reference env file)
            cfg.seed = seed
            cfg.dtype = L.F64 if dtype in ("f64", torch.float64) else L.F32
            cfg.flags = (L.FLAG_AUTO_RESET if auto_reset else 0) | (L.FLAG_NAN_GUARD if nan_guard else 0)
            cfg.env_id_offset = env_id_offset
            cfg.block_size = block_size
            cfg.step_kernel = L.KERNELS[kernel]   # "auto" | "lane" | "helper" | "team": which implementation of the same step runs
            if ee_task is not None:               # arm vehicles: "tool" (default) | "base" -- the point the waypoint task measures from
                cfg.task.ee_task = {"base": L.EE_TASK_BASE, "tool": L.EE_TASK_TOOL}[ee_task]
            cfg.task.rk4_substeps = rk4_substeps
            if max_episode_steps is not None:
                cfg.task.max_episode_steps = max_episode_steps
            if counter_limit is not None:
                cfg.task.counter_limit = counter_limit
            if num_waypoints != 1 and task == "v2":
                import math
                cfg.task.num_waypoints = num_waypoints
                for k in range(1, num_waypoints + 1):
                    t = k / num_waypoints
                    cfg.task.traj_sin[k - 1] = math.sin(2.0 * t * math.pi)
                    cfg.task.traj_cos[k - 1] = math.cos(t * 2.0 * math.pi)
        else:
            cfg = config
        self.cfg = cfg
        self.num_envs = cfg.num_envs
        od, ad, nf, ni = (C.c_int32() for _ in range(4))
        self.lib.amenv_dims(C.byref(cfg), C.byref(od), C.byref(ad), C.byref(nf), C.byref(ni))
        self.obs_dim, self.act_dim, self.n_float_fields, self.n_int_fields = od.value, ad.value, nf.value, ni.value
        self.state_dtype = torch.float64 if cfg.dtype == L.F64 else torch.float32
        self.bytes_per_env_step = int(self.lib.amenv_bytes_per_env_step(C.byref(cfg)))
        h = C.c_void_p()
        rc = self.lib.amenv_create(C.byref(cfg), self.device_index, C.byref(h))
        if rc != 0:
            raise L.AmenvError(f"amenv_create failed ({rc}): {self.lib.amenv_last_error(None).decode()}")
        self._h = h
        n, dev = self.num_envs, self.device
        self.obs = torch.zeros(n, self.obs_dim, dtype=torch.float32, device=dev)
        self.reward = torch.zeros(n, dtype=self.state_dtype, device=dev)
        self.done = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.info_bits = torch.zeros(n, dtype=torch.int32, device=dev)
        self.terminal_obs = torch.zeros(n, self.obs_dim, dtype=torch.float32, device=dev)
        self.ep_return = torch.zeros(n, dtype=torch.float32, device=dev)
        self.ep_len = torch.zeros(n, dtype=torch.int32, device=dev)
        self.kernel_name = self.lib.amenv_kernel_name(self._h).decode()

    # ------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, rc, what):
        if rc != 0:
            raise L.AmenvError(f"{what} failed ({rc}): {self.lib.amenv_last_error(self._h).decode()}")

    def _actions(self, actions, lead=()):
        want = tuple(lead) + (self.num_envs, self.act_dim)
        if actions.device != self.device or actions.dtype != torch.float32 or tuple(actions.shape) != want or not actions.is_contiguous():
            actions = actions.to(device=self.device, dtype=torch.float32).reshape(want).contiguous()
        if actions.data_ptr() % 16:    # e.g. a row of a [T, N, A] tensor with odd N*A: the kernels want 16-byte aligned buffers
            actions = actions.clone()
        return actions

    def reseed(self, seed):
        """Re-key the reset RNG for episodes started from now on."""
        self._check(self.lib.amenv_set_seed(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF), "amenv_set_seed")
        self.cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF

    def reset(self, mask=None):
        """WaypointQuadEnv.reset (v2/rl_env_scaledObs.py:40-79) for all envs, or those with mask != 0."""
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        self._check(self.lib.amenv_reset(self._h, None if m is None else C.c_void_p(m.data_ptr()), C.c_void_p(self.obs.data_ptr()),
                                         self._stream()), "amenv_reset")
        return self.obs

    def step(self, actions):
        """WaypointQuadEnv.step (v2/rl_env_scaledObs.py:123-196) for all N envs: one kernel launch."""
        a = self._actions(actions)
        self._check(self.lib.amenv_step(self._h, C.c_void_p(a.data_ptr()), C.c_void_p(self.obs.data_ptr()),
                                        C.c_void_p(self.reward.data_ptr()), C.c_void_p(self.done.data_ptr()),
                                        C.c_void_p(self.info_bits.data_ptr()), C.c_void_p(self.terminal_obs.data_ptr()),
                                        C.c_void_p(self.ep_return.data_ptr()), C.c_void_p(self.ep_len.data_ptr()), self._stream()),
                    "amenv_step")
        return self.obs, self.reward, self.done, self.info_bits

    def step_into(self, actions, obs, reward, done):
        """step() writing obs [N,obs_dim] f32, reward [N], done [N] u8 straight into caller tensors (e.g. rows of a rollout
        buffer) instead of the persistent output buffers; info_bits / terminal_obs / ep_return / ep_len as in step()."""
        a = self._actions(actions)
        n = self.num_envs
        ok = (obs.device == self.device and obs.dtype == torch.float32 and obs.is_contiguous() and tuple(obs.shape) == (n, self.obs_dim)
              and reward.device == self.device and reward.dtype == self.state_dtype and reward.is_contiguous() and reward.numel() == n
              and done.device == self.device and done.dtype == torch.uint8 and done.is_contiguous() and done.numel() == n)
        if not ok:
            raise L.AmenvError("step_into: obs/reward/done must be contiguous tensors of this env's device, dtype and shape")
        self._check(self.lib.amenv_step(self._h, C.c_void_p(a.data_ptr()), C.c_void_p(obs.data_ptr()),
                                        C.c_void_p(reward.data_ptr()), C.c_void_p(done.data_ptr()),
                                        C.c_void_p(self.info_bits.data_ptr()), C.c_void_p(self.terminal_obs.data_ptr()),
                                        C.c_void_p(self.ep_return.data_ptr()), C.c_void_p(self.ep_len.data_ptr()), self._stream()),
                    "amenv_step")
        return obs, reward, done, self.info_bits

    def step_timed(self, actions):
        """step() that also returns the device-side duration (us) of that one kernel launch (synchronises)."""
        a = self._actions(actions)
        us = C.c_float()
        self._check(self.lib.amenv_step_timed(self._h, C.c_void_p(a.data_ptr()), C.c_void_p(self.obs.data_ptr()),
                                              C.c_void_p(self.reward.data_ptr()), C.c_void_p(self.done.data_ptr()),
                                              C.c_void_p(self.info_bits.data_ptr()), C.c_void_p(self.terminal_obs.data_ptr()),
                                              C.c_void_p(self.ep_return.data_ptr()), C.c_void_p(self.ep_len.data_ptr()), self._stream(),
                                              C.byref(us)), "amenv_step_timed")
        return us.value

    def rollout(self, actions, want_obs=True, want_flags=True):
        """T open-loop steps in one launch; actions [T,N,4].  Returns dict of [T,N,...] tensors."""
        T = int(actions.shape[0])
        a = self._actions(actions, lead=(T,))
        n, dev = self.num_envs, self.device
        out = dict(reward=torch.empty(T, n, dtype=self.state_dtype, device=dev))
        if want_obs:
            out["obs"] = torch.empty(T, n, self.obs_dim, dtype=torch.float32, device=dev)
        if want_flags:
            out["done"] = torch.empty(T, n, dtype=torch.uint8, device=dev)
            out["info_bits"] = torch.empty(T, n, dtype=torch.int32, device=dev)
        p = lambda k: C.c_void_p(out[k].data_ptr()) if k in out else None
        self._check(self.lib.amenv_rollout(self._h, T, C.c_void_p(a.data_ptr()), p("obs"), p("reward"), p("done"), p("info_bits"),
                                           self._stream()), "amenv_rollout")
        return out

    def rollout_policy(self, flat_params, n_steps, seed, draw0, obs, actions, logp, values, rewards, dones, info_bits=None, terminal_obs=None):
        """T closed-loop steps in ONE launch: obs -> actor / critic MLPs (bf16 matrix cores) -> Gaussian sample -> clip -> env step,
        writing SB3's rollout-buffer rows: obs [T+1,N,OD] (row 0 = observation at entry), actions [T,N,A] raw samples, logp / values /
        rewards [T,N] f32, dones [T,N] u8, optionally info_bits [T,N] i32 and terminal_obs [T,N,OD].  Caller-owned contiguous tensors on
        this env's device.  Built for the fp32 hexacopter + arm (include/amenv.h amenv_rollout_policy)."""
        T, n = int(n_steps), self.num_envs
        want = {"obs": ((T + 1, n, self.obs_dim), torch.float32), "actions": ((T, n, self.act_dim), torch.float32), "logp": ((T, n), torch.float32),
                "values": ((T, n), torch.float32), "rewards": ((T, n), torch.float32), "dones": ((T, n), torch.uint8)}
        got = {"obs": obs, "actions": actions, "logp": logp, "values": values, "rewards": rewards, "dones": dones}
        if info_bits is not None:      # optional outputs are written [T,N] / [T,N,OD] by the kernel all the same: check them like the rest
            want["info_bits"] = ((T, n), torch.int32); got["info_bits"] = info_bits
        if terminal_obs is not None:
            want["terminal_obs"] = ((T, n, self.obs_dim), torch.float32); got["terminal_obs"] = terminal_obs
        for k, (shape, dt) in want.items():
            t = got[k]
            if t.device != self.device or t.dtype != dt or tuple(t.shape) != shape or not t.is_contiguous():
                raise L.AmenvError(f"rollout_policy: {k} must be a contiguous {dt} tensor of shape {shape} on {self.device}")
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        fp = flat_params.detach()
        if fp.device != self.device or fp.dtype != torch.float32 or not fp.is_contiguous():
            raise L.AmenvError("rollout_policy: flat_params must be a contiguous fp32 tensor on this env's device")
        self._check(self.lib.amenv_rollout_policy(self._h, T, p(fp), int(seed) & 0xFFFFFFFFFFFFFFFF, int(draw0) & 0xFFFFFFFF, p(obs), p(actions), p(logp),
                                                  p(values), p(rewards), p(dones), p(info_bits), p(terminal_obs), self._stream()), "amenv_rollout_policy")

    def observe(self):
        o = torch.empty(self.num_envs, self.obs_dim, dtype=torch.float32, device=self.device)
        self._check(self.lib.amenv_observe(self._h, C.c_void_p(o.data_ptr()), self._stream()), "amenv_observe")
        return o

    def ee_position(self):
        """World position [N,3] f32 of the arm's tool point (forward kinematics of the current state; the body origin without an arm)."""
        o = torch.empty(self.num_envs, 3, dtype=torch.float32, device=self.device)
        self._check(self.lib.amenv_ee_position(self._h, C.c_void_p(o.data_ptr()), self._stream()), "amenv_ee_position")
        return o

    def get_state(self):
        """(fstate [n_float_fields,N] of the state dtype, istate [4,N] int32): the SoA episode state."""
        f = torch.empty(self.n_float_fields, self.num_envs, dtype=self.state_dtype, device=self.device)
        i = torch.empty(self.n_int_fields, self.num_envs, dtype=torch.int32, device=self.device)
        self._check(self.lib.amenv_get_state(self._h, C.c_void_p(f.data_ptr()), C.c_void_p(i.data_ptr()), self._stream()), "amenv_get_state")
        return f, i

    def set_state(self, fstate=None, istate=None):
        f = i = None
        if fstate is not None:
            f = torch.as_tensor(fstate).to(device=self.device, dtype=self.state_dtype).contiguous()
            assert tuple(f.shape) == (self.n_float_fields, self.num_envs), f.shape
        if istate is not None:
            i = torch.as_tensor(istate).to(device=self.device, dtype=torch.int32).contiguous()
            assert tuple(i.shape) == (self.n_int_fields, self.num_envs), i.shape
        self._check(self.lib.amenv_set_state(self._h, None if f is None else C.c_void_p(f.data_ptr()),
                                             None if i is None else C.c_void_p(i.data_ptr()), self._stream()), "amenv_set_state")
        # keep the staging tensors alive until the copy has been enqueued AND executed
        torch.cuda.current_stream(self.device).synchronize()

    def stats(self, reset=False):
        s = L.Stats()
        self._check(self.lib.amenv_stats_read(self._h, C.byref(s), 1 if reset else 0, self._stream()), "amenv_stats_read")
        d = {k: int(getattr(s, k)) for k, _ in L.Stats._fields_}
        d["return_sum"] = d.pop("return_sum_q10") / 1024.0
        return d

    def close(self):
        if getattr(self, "_h", None):
            self.lib.amenv_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
