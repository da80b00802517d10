"""Observation normalisation on the GPU: the equivalent of `VecNormalize(env, norm_obs=True, norm_reward=False)`
that v1/rl_train_vecN.py:10-11 and v1/rl_checkpoint_train_vecN.py:19-26 wrap the env in.

`ObsNormalizer`   -- running mean/var/count of [n, dim] f32 device batches + normalisation (HIP kernels behind
                     amenv_obsnorm_* in include/amenv.h), for GPU-resident loops.
`GpuVecNormalize` -- SB3 `VecNormalize`-shaped wrapper over GpuVecEnv (numpy API): normalises the observations returned
                     by reset()/step_wait() and the `terminal_observation` in infos, `training` flag, `save`/`load`.
Statistics are saved as .npz (mean, var, count, clip_obs, epsilon).  The reference's `vec_normalize.pkl` is a pickle of
an SB3 object and is deliberately not read (no unpickling of reference artefacts).  SB3 2.6.0 semantics; parity unpinned.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


class ObsNormalizer:
    def __init__(self, dim, device=0, clip_obs=10.0, epsilon=1e-8):
        self.lib = L.load()
        self.dim, self.clip_obs, self.epsilon = int(dim), float(clip_obs), float(epsilon)
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        h = C.c_void_p()
        rc = self.lib.amenv_obsnorm_create(self.dim, self.device.index, C.byref(h))
        if rc != 0:
            raise L.AmenvError(f"amenv_obsnorm_create failed ({rc}): {self.lib.amenv_last_error(None).decode()}")
        self._h = h

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, obs):
        t = torch.as_tensor(obs)
        if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous():
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
        assert t.shape[-1] == self.dim, t.shape
        return t

    def update(self, obs):
        """obs_rms.update(obs): merge this batch's mean / population variance / count."""
        t = self._dev(obs)
        rc = self.lib.amenv_obsnorm_update(self._h, C.c_void_p(t.data_ptr()), t.numel() // self.dim, self._stream())
        if rc != 0:
            raise L.AmenvError(f"amenv_obsnorm_update failed ({rc})")

    def normalize(self, obs, out=None):
        """clip((obs - mean) / sqrt(var + epsilon), -clip_obs, clip_obs) -> new tensor (or `out`, may alias obs)."""
        t = self._dev(obs)
        o = torch.empty_like(t) if out is None else out
        rc = self.lib.amenv_obsnorm_apply(self._h, C.c_void_p(t.data_ptr()), C.c_void_p(o.data_ptr()), t.numel() // self.dim,
                                          self.clip_obs, self.epsilon, self._stream())
        if rc != 0:
            raise L.AmenvError(f"amenv_obsnorm_apply failed ({rc})")
        return o

    def __call__(self, obs, update=True):
        if update:
            self.update(obs)
        return self.normalize(obs)

    def get(self):
        mean, var, count = np.zeros(self.dim), np.zeros(self.dim), np.zeros(1)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        if self.lib.amenv_obsnorm_get(self._h, p(mean), p(var), p(count), self._stream()) != 0:
            raise L.AmenvError("amenv_obsnorm_get failed")
        return mean, var, float(count[0])

    def set(self, mean, var, count):
        mean, var = np.ascontiguousarray(mean, np.float64), np.ascontiguousarray(var, np.float64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        if self.lib.amenv_obsnorm_set(self._h, p(mean), p(var), float(count), self._stream()) != 0:
            raise L.AmenvError("amenv_obsnorm_set failed")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.amenv_obsnorm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GpuVecNormalize:
    """`VecNormalize(venv, norm_obs=True, norm_reward=False)` over a GpuVecEnv (same method names as SB3's wrapper)."""

    def __init__(self, venv, training=True, norm_obs=True, norm_reward=False, clip_obs=10.0, epsilon=1e-8, normalizer=None):
        if norm_reward:
            raise NotImplementedError("the reference uses norm_reward=False (v1/rl_train_vecN.py:11)")
        self.venv, self.training, self.norm_obs = venv, training, norm_obs
        self.num_envs, self.observation_space, self.action_space = venv.num_envs, venv.observation_space, venv.action_space
        dim = int(self.observation_space.shape[0])
        self.obs_rms = normalizer if normalizer is not None else ObsNormalizer(dim, getattr(getattr(venv, "backend", None), "device_index", 0), clip_obs, epsilon)
        self.clip_obs, self.epsilon = clip_obs, epsilon
        self.old_obs = None

    def _norm(self, obs_np, update):
        self.old_obs = obs_np
        if not self.norm_obs:
            return obs_np
        t = torch.from_numpy(np.ascontiguousarray(obs_np, np.float32))
        if update and self.training:
            self.obs_rms.update(t)
        return self.obs_rms.normalize(t).cpu().numpy()

    def reset(self):
        return self._norm(self.venv.reset(), True)

    def step_async(self, actions):
        self.venv.step_async(actions)

    def step_wait(self):
        obs, rew, done, infos = self.venv.step_wait()
        obs = self._norm(obs, True)
        if self.norm_obs:
            idx = [i for i in np.nonzero(done)[0] if "terminal_observation" in infos[i]]
            if idx:
                t = self.obs_rms.normalize(torch.from_numpy(np.stack([infos[i]["terminal_observation"] for i in idx]).astype(np.float32))).cpu().numpy()
                for k, i in enumerate(idx):
                    infos[i]["terminal_observation"] = t[k]
        return obs, rew, done, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def normalize_obs(self, obs):
        return self.obs_rms.normalize(torch.from_numpy(np.ascontiguousarray(obs, np.float32))).cpu().numpy()

    def get_original_obs(self):
        return self.old_obs

    def save(self, path):
        mean, var, count = self.obs_rms.get()
        np.savez(path, mean=mean, var=var, count=count, clip_obs=self.clip_obs, epsilon=self.epsilon)

    @classmethod
    def load(cls, path, venv):
        d = np.load(path if str(path).endswith(".npz") else str(path) + ".npz")
        self = cls(venv, clip_obs=float(d["clip_obs"]), epsilon=float(d["epsilon"]))
        self.obs_rms.set(d["mean"], d["var"], float(d["count"]))
        return self

    def close(self):
        self.obs_rms.close()
        self.venv.close()

    def __getattr__(self, name):  # seed, get_attr, env_method, ... fall through to the wrapped env
        return getattr(self.venv, name)
