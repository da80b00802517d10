"""BASELINE config 5: PPO rollout + update with the environment, the rollout buffer and the policy resident on the GPU.

    python tools/ppo_bench.py [--envs 32768] [--n-steps 32] [--batch 65536] [--epochs 12] [--iters 3] [--vehicle quad]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/ppo_bench.py ...

One process per GPU, envs sharded by global id; the only collective is one all-reduce of the flat gradient buffer per
minibatch (RCCL).  Prints ONE JSON line on rank 0: whole-job env-steps/s through the full PPO loop (rollout + GAE + update),
with the rollout and update phases timed separately (max over ranks).  This is the reference's `time/fps` quantity
(139-391 env-steps/s with 8 CPU envs), not the headline metric of bench.py.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=32768, help="environments per GPU")
    ap.add_argument("--n-steps", type=int, default=32)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--epochs", type=int, default=12)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--vehicle", default="quad")
    ap.add_argument("--fused-rollout", action="store_true", help="collect_rollouts as one launch of amenv_rollout_policy (hexa_arm)")
    a = ap.parse_args()
    import torch
    import rl_aerial_manipulator_amd as amd
    from rl_aerial_manipulator_amd import sharding
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    sh = sharding.shard_from_env(a.envs)
    dist = sharding.init_process_group("nccl", dev)
    env = amd.GpuWaypointEnv(a.envs, device=local, vehicle=a.vehicle, seed=0, env_id_offset=sh.env_id_offset)
    algo = amd.PPO(env, n_steps=a.n_steps, batch_size=a.batch, n_epochs=a.epochs, seed=0, dist=dist, fused_rollout=a.fused_rollout)
    t_roll = t_upd = 0.0
    rec = {}
    for it in range(a.warmup + a.iters):
        if it == a.warmup:
            torch.cuda.synchronize(); sharding.barrier(dist)
            t_roll = t_upd = 0.0
            t_all = time.perf_counter()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        algo.collect_rollouts()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        rec = algo.train()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        t_roll += t1 - t0; t_upd += t2 - t1
    sharding.barrier(dist)
    total = sharding.max_over_ranks(dist, time.perf_counter() - t_all, dev)
    t_roll = sharding.max_over_ranks(dist, t_roll, dev)
    t_upd = sharding.max_over_ranks(dist, t_upd, dev)
    samples = a.iters * a.n_steps * a.envs * sh.world
    if sh.rank == 0:
        print(json.dumps({
            "metric": "PPO env-steps/sec (rollout + GAE + update), whole job", "value": samples / total, "unit": "env-steps/s",
            "n_gpus": sh.world, "iters": a.iters, "rollout_s_per_iter": t_roll / a.iters, "update_s_per_iter": t_upd / a.iters,
            "rollout_env_steps_per_s": samples / t_roll, "minibatches_per_iter": a.epochs * -(-a.n_steps * a.envs // a.batch),
            "config": {"workload": f"{a.vehicle} PPO MLP[128,64,64] tanh", "envs_per_gpu": a.envs, "n_steps": a.n_steps, "batch_size": a.batch,
                       "n_epochs": a.epochs, "parameters": algo.policy.num_parameters(), "grad_allreduce_bytes": 4 * algo.policy.num_parameters()},
            "dtype": "f32", "last_losses": rec}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
