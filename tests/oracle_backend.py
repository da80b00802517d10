"""Test-only stand-in for GpuWaypointEnv built on the CPU oracle, so the adapters' host logic (info dicts,
auto-reset bookkeeping, attribute surface) can be exercised without a GPU.  Never imported by the product."""
import numpy as np
import torch

from oracle import oracle as O


class OracleBackend:
    def __init__(self, num_envs, seed=0, auto_reset=True, num_waypoints=1, max_episode_steps=None, variant=O.TASK_V2_SCALED20):
        self.cfg = O.reference_quad_config(num_envs=num_envs, seed=seed, flags=O.FLAG_AUTO_RESET if auto_reset else 0,
                                           num_waypoints=num_waypoints, variant=variant)
        if max_episode_steps is not None:
            self.cfg.task.max_episode_steps = max_episode_steps
        self.env = O.OracleEnv(self.cfg)
        self.num_envs, self.obs_dim, self.act_dim = num_envs, self.env.obs_dim, 4
        self.terminal_obs = torch.zeros(num_envs, self.obs_dim)
        self.ep_return = torch.zeros(num_envs)
        self.ep_len = torch.zeros(num_envs, dtype=torch.int32)

    def reseed(self, seed):
        self.cfg.seed = seed

    def reset(self, mask=None):
        return torch.from_numpy(self.env.reset(None if mask is None else np.asarray(mask)))

    def observe(self):
        return torch.from_numpy(self.env.observe())

    def step(self, actions):
        o = self.env.step(np.asarray(actions, np.float32))
        d = o["done"].astype(bool)
        self.terminal_obs[d] = torch.from_numpy(o["terminal_obs"][d])
        self.ep_return[d] = torch.from_numpy(o["ep_return"][d])
        self.ep_len[d] = torch.from_numpy(o["ep_len"][d])
        return (torch.from_numpy(o["obs"]), torch.from_numpy(o["reward"]), torch.from_numpy(o["done"]),
                torch.from_numpy(o["info"].view(np.int32)))

    def get_state(self):
        return torch.from_numpy(self.env.fstate.copy()), torch.from_numpy(self.env.istate.copy())

    def set_state(self, f=None, i=None):
        if f is not None:
            self.env.fstate[:] = np.asarray(f)
        if i is not None:
            self.env.istate[:] = np.asarray(i)

    def close(self):
        pass
