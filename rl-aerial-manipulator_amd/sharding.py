"""Multi-GPU host logic: the environment batch shards embarrassingly over ranks.

One process per GPU.  Rank r owns global envs [r*n, (r+1)*n): it creates its GpuWaypointEnv with
`env_id_offset = r*n`, and because the reset RNG is keyed by GLOBAL env id the union of the shards is
bit-identical to one big environment (tests/test_gpu_parity.py::test_sharding_invariance on the GPU,
tests/test_distributed_cpu.py under gloo).  There is NO collective on the step path; torch.distributed
(RCCL on GPUs, gloo on CPU) is used only around it: barrier + max-over-ranks of the timing, and an
optional sum of the episode counters for logging.
"""
import os
from dataclasses import dataclass


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    envs_per_rank: int

    @property
    def env_id_offset(self):
        return self.rank * self.envs_per_rank

    @property
    def global_envs(self):
        return self.world * self.envs_per_rank

    @property
    def global_ids(self):
        return range(self.env_id_offset, self.env_id_offset + self.envs_per_rank)


def shard_from_env(envs_per_rank):
    """Shard of this process from the torchrun environment (RANK / WORLD_SIZE); single process if unset."""
    return Shard(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(envs_per_rank))


def init_process_group(backend, device=None):
    """Join the job's process group (no-op for a single process).  Rendezvous on 127.0.0.1 unless told otherwise."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, **kw)
    return dist


def barrier(dist):
    if dist is not None:
        dist.barrier()


def max_over_ranks(dist, seconds, device="cpu"):
    """The job's step time is the slowest rank's."""
    if dist is None:
        return float(seconds)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_counters(dist, counters, device="cpu"):
    """Sum a dict of integer episode counters over ranks (logging only; never on the step path)."""
    if dist is None:
        return dict(counters)
    import torch
    keys = sorted(counters)
    t = torch.tensor([int(counters[k]) for k in keys], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {k: int(v) for k, v in zip(keys, t.tolist())}


def whole_job_rate(shard, steps, seconds):
    """env-steps/s of the WHOLE job: every rank stepped its n envs `steps` times in `seconds` (max over ranks)."""
    return shard.global_envs * steps / seconds
