"""GpuVecEnv -- Stable-Baselines3 `VecEnv`-shaped adapter over the GPU-resident environment.

Replaces, for the reference's training entry points, what SB3 builds at
`make_vec_env(WaypointQuadEnv, n_envs=8)` (v2/rl_train.py:24): `DummyVecEnv([Monitor(WaypointQuadEnv())]*8)`,
a sequential Python loop over single envs.  Here all N envs step in one HIP kernel launch; this class only
converts at the numpy boundary SB3 needs and rebuilds the per-env `info` dicts:

  reference info dict keys (v2/rl_env_scaledObs.py:164,173,179,192,195):  success, stopped, crashed, out_of_bounds
  DummyVecEnv on done:   infos[i]["terminal_observation"], obs[i] <- reset obs          (SB3 2.6.0)
  TimeLimit bookkeeping: infos[i]["TimeLimit.truncated"] = truncated and not terminated
  Monitor on done:       infos[i]["episode"] = {"r": return, "l": length, "t": seconds since start}

Drop-in: `make_vec_env(WaypointQuadEnv, n_envs=N, vec_env_cls=GpuVecEnv)` -- the `env_fns` SB3 passes are counted,
not called.  SB3 / gymnasium are optional imports: with them installed this class subclasses SB3's `VecEnv` and uses
gymnasium `Box` spaces; without them it is a structural duck-type (the parity tests use local stubs).
SB3 and gymnasium are third-party and un-pinned in the reference (v2/requirements.txt:4-5): the auto-reset /
Monitor behaviour above follows SB3 2.6.0's documented semantics -- parity unpinned (no reference test covers it).
"""
import time

import numpy as np

from . import _lib as L

try:  # optional third-party bases
    from stable_baselines3.common.vec_env import VecEnv as _SB3VecEnv
except Exception:  # noqa: BLE001 - absent or broken install: structural duck-type
    _SB3VecEnv = object
try:
    from gymnasium import spaces as _spaces
except Exception:  # noqa: BLE001
    _spaces = None


class Box:
    """Minimal stand-in for gymnasium.spaces.Box when gymnasium is not installed."""

    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape) if shape is not None else np.shape(low)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape).copy()

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return np.random.uniform(lo, hi).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


def make_spaces(obs_dim=20, act_dim=4):
    """observation_space / action_space of WaypointQuadEnv (v2/rl_env_scaledObs.py:14-24; 17-D for the v1 envs,
    v1/rl_env_scaledObs.py:13-23)."""
    B = _spaces.Box if _spaces is not None else Box
    obs = B(low=-np.inf, high=np.inf, shape=(int(obs_dim),), dtype=np.float32)
    # thrust in [0, 2], moments (and, with the arm, joint position commands) in [-1, 1]
    lo = np.array([0.0] + [-1.0] * (int(act_dim) - 1), dtype=np.float32)
    hi = np.array([2.0] + [1.0] * (int(act_dim) - 1), dtype=np.float32)
    act = B(low=lo, high=hi, dtype=np.float32)
    return obs, act


def info_dict(bits):
    """The reference's `info` dict for one env from its info bits (v2/rl_env_scaledObs.py:164-196)."""
    d = {}
    if bits & L.INFO_SUCCESS:
        d["success"] = True
        d["stopped"] = bool(bits & L.INFO_STOPPED)
    elif bits & L.INFO_CRASHED:
        d["success"] = False
        d["crashed"] = True
    elif bits & L.INFO_OOB:
        d["success"] = False
        d["out_of_bounds"] = True
    if bits & L.INFO_NONFINITE:
        d["success"] = False
        d["nonfinite"] = True
    return d


class GpuVecEnv(_SB3VecEnv):
    """N GPU-resident WaypointQuadEnv behind SB3's VecEnv interface (numpy in / numpy out)."""

    def __init__(self, env_fns=None, num_envs=None, device=0, backend=None, **env_kwargs):
        if backend is None:
            n = num_envs if num_envs is not None else (len(env_fns) if env_fns is not None else None)
            if n is None:
                raise ValueError("GpuVecEnv needs env_fns (as SB3's make_vec_env passes) or num_envs")
            from .gpu_env import GpuWaypointEnv
            backend = GpuWaypointEnv(int(n), device=device, auto_reset=True, **env_kwargs)
        self.backend = backend
        obs_space, act_space = make_spaces(getattr(backend, "obs_dim", 20), getattr(backend, "act_dim", 4))
        if _SB3VecEnv is not object:
            super().__init__(backend.num_envs, obs_space, act_space)
        else:
            self.num_envs, self.observation_space, self.action_space = backend.num_envs, obs_space, act_space
        self.render_mode = None
        self._actions = None
        self._t0 = time.time()
        self.dt = 1.0 / 200.0  # v2/rl_env_scaledObs.py:30
        self._pin = self._make_staging(backend)

    # ---- VecEnv API ---------------------------------------------------------------------------
    def reset(self):
        self._t0 = time.time()
        return self._to_numpy(self.backend.reset())

    def step_async(self, actions):
        self._actions = np.ascontiguousarray(actions, dtype=np.float32).reshape(self.num_envs, -1)

    def step_wait(self):
        import torch
        b = self.backend
        if self._pin is None:   # host-resident backend (the oracle stand-in of the CPU tests)
            obs, rew, done, bits = b.step(torch.from_numpy(self._actions))
            obs, rew = self._to_numpy(obs), self._to_numpy(rew).astype(np.float32)
            done, bits = self._to_numpy(done).astype(bool), self._to_numpy(bits).view(np.uint32)
        else:
            # numpy boundary through page-locked staging: one async H2D copy of the actions, the step launch, six async D2H copies
            # (Monitor's return / length ride along: 8 B per env, so that an episode end costs no synchronisation of its own),
            # ONE stream synchronisation (pageable tensors made each of the copies a blocking staged transfer)
            P = self._pin
            np.copyto(P["act_np"], self._actions)
            P["act_dev"].copy_(P["act"], non_blocking=True)
            o, r, d, i = b.step(P["act_dev"])
            P["obs"].copy_(o, non_blocking=True); P["rew"].copy_(r, non_blocking=True)
            P["done"].copy_(d, non_blocking=True); P["bits"].copy_(i, non_blocking=True)
            P["eret"].copy_(b.ep_return, non_blocking=True); P["elen"].copy_(b.ep_len, non_blocking=True)
            torch.cuda.current_stream(b.device).synchronize()
            # SB3 keeps `_last_obs` across the next step: hand out copies, never views of the staging buffers
            obs, rew = P["obs_np"].copy(), P["rew_np"].astype(np.float32)
            done, bits = P["done_np"].astype(bool), P["bits_np"].view(np.uint32).copy()
        infos = [{} for _ in range(self.num_envs)]   # one dict PER env: SB3 wrappers write into infos[i]
        ev = np.nonzero(bits & ~np.uint32(L.INFO_WAS_RESET))[0]
        if ev.size:
            dn = ev[done[ev]]
            if self._pin is not None:   # the terminal rows of the few envs that ended: one gathered copy; return / length are staged already
                tobs = self._to_numpy(b.terminal_obs[torch.from_numpy(dn).to(b.device)]) if dn.size else None
                eret, elen = self._pin["eret_np"][dn], self._pin["elen_np"][dn]
            else:
                tobs = self._to_numpy(b.terminal_obs[dn]) if dn.size else None
                eret = self._to_numpy(b.ep_return[dn]) if dn.size else None
                elen = self._to_numpy(b.ep_len[dn]) if dn.size else None
            pos = {int(i): k for k, i in enumerate(dn)}
            now = round(time.time() - self._t0, 6)
            for i in ev:
                i = int(i)
                d = info_dict(int(bits[i]))
                if done[i]:
                    k = pos[i]
                    d["terminal_observation"] = tobs[k]
                    d["TimeLimit.truncated"] = bool(bits[i] & L.INFO_TRUNCATED) and not bool(bits[i] & L.INFO_TERMINATED)
                    d["episode"] = {"r": float(eret[k]), "l": int(elen[k]), "t": now}
                infos[i] = d
        return obs, rew, done, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        if hasattr(self.backend, "close"):
            self.backend.close()

    def seed(self, seed=None):
        """SB3 calls this before the first reset.  The reset RNG is counter-based (seed, env id, episode):
        re-seeding takes effect for episodes started from now on."""
        if seed is not None and hasattr(self.backend, "reseed"):
            self.backend.reseed(int(seed))
        return [seed] * self.num_envs

    def get_attr(self, attr_name, indices=None):
        idx = self._indices(indices)
        if attr_name == "render_mode":
            return [None for _ in idx]
        return [getattr(self, attr_name) for _ in idx]

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        return [getattr(self, method_name)(*method_args, **method_kwargs) for _ in self._indices(indices)]

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False for _ in self._indices(indices)]

    def get_images(self):
        return [None] * self.num_envs

    def render(self, mode=None):
        return None

    # ---- helpers --------------------------------------------------------------------------------
    def _indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        return [indices] if isinstance(indices, int) else list(indices)

    @staticmethod
    def _make_staging(backend):
        """Page-locked host mirrors of the step's inputs / outputs when the backend lives on a GPU; None otherwise."""
        dev = getattr(backend, "device", None)
        if dev is None or getattr(dev, "type", "cpu") != "cuda":
            return None
        import torch
        n, od, ad = backend.num_envs, backend.obs_dim, backend.act_dim
        P = {"act": torch.empty(n, ad, dtype=torch.float32).pin_memory(), "act_dev": torch.empty(n, ad, dtype=torch.float32, device=dev),
             "obs": torch.empty(n, od, dtype=torch.float32).pin_memory(), "rew": torch.empty(n, dtype=backend.reward.dtype).pin_memory(),
             "done": torch.empty(n, dtype=torch.uint8).pin_memory(), "bits": torch.empty(n, dtype=torch.int32).pin_memory(),
             "eret": torch.empty(n, dtype=backend.ep_return.dtype).pin_memory(), "elen": torch.empty(n, dtype=backend.ep_len.dtype).pin_memory()}
        for k in ("act", "obs", "rew", "done", "bits", "eret", "elen"):
            P[k + "_np"] = P[k].numpy()
        return P

    @staticmethod
    def _to_numpy(t):
        return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
