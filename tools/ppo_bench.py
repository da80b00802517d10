"""BASELINE config 5: PPO rollout + update with the environment, the rollout buffer and the policy resident on the GPU.

    python tools/ppo_bench.py [--gpus N] [--envs 32768] [--n-steps 32] [--batch 65536] [--epochs 12] [--iters 3] [--vehicle quad]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/ppo_bench.py ...   (the same thing, started by hand)

`--gpus N` without a torchrun environment starts the N ranks itself as CHILD processes -- before this process imports torch or touches a GPU,
and without exec -- and relays rank 0's line (bench.py's launcher; BASELINE configs[4] is `--gpus 8 --envs 32768`).  `--dry-run` replaces the
env and the learner by a no-op so that launcher, process group (gloo on CPU), barriers and max-over-ranks timing can be checked without a GPU.

One process per GPU, envs sharded by global id; the only collective is one all-reduce of the flat gradient buffer per
minibatch (RCCL).  Prints ONE JSON line on rank 0: whole-job env-steps/s through the full PPO loop (rollout + GAE + update),
with the rollout and update phases timed separately (max over ranks).  This is the reference's `time/fps` quantity
(139-391 env-steps/s with 8 CPU envs), not the headline metric of bench.py.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def launch_children(gpus):
    """Parent of an N-rank job.  Runs before torch is imported (nothing here has initialised the GPU); never exec()s."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in p.stdout.splitlines():
        try:
            if "metric" in json.loads(ln):
                line = ln
        except ValueError:
            sys.stderr.write(ln + "\n")
    if p.returncode != 0 or line is None:
        sys.stderr.write(f"ppo_bench.py: the {gpus}-rank job failed (exit {p.returncode})\n")
        return p.returncode or 1
    print(line, flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks (one per GPU); > 1 without a torchrun environment: this process starts them as children")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) | gloo (CPU tests)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / process-group / timing plumbing only (no env, no learner): CPU test of --gpus N")
    ap.add_argument("--envs", type=int, default=32768, help="environments per GPU")
    ap.add_argument("--n-steps", type=int, default=32)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--epochs", type=int, default=12)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--vehicle", default="quad")
    ap.add_argument("--fused-rollout", action="store_true", help="collect_rollouts as one launch of amenv_rollout_policy (hexa_arm)")
    a = ap.parse_args()
    if a.gpus > 1 and "RANK" not in os.environ:
        return launch_children(a.gpus)
    import torch
    import rl_aerial_manipulator_amd as amd
    from rl_aerial_manipulator_amd import sharding
    local = int(os.environ.get("LOCAL_RANK", "0"))
    sh = sharding.shard_from_env(a.envs)
    if a.dry_run:
        dev = torch.device("cpu")
        dist = sharding.init_process_group(a.backend, None)

        class _Dry:   # no-op stand-ins: what is exercised is everything around them
            class policy:
                @staticmethod
                def num_parameters():
                    return 30537
            def collect_rollouts(self): pass
            def train(self): return {}
        algo = _Dry()
        torch.cuda.synchronize = lambda *x: None
    else:
        if local >= torch.cuda.device_count():   # counting devices does not initialise the GPU
            raise SystemExit(f"ppo_bench.py: rank {sh.rank} wants GPU {local} but only {torch.cuda.device_count()} are visible (--gpus too large)")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        dist = sharding.init_process_group(a.backend, dev)
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
            "n_gpus": sh.world, "rccl_ranks": sh.world if dist is not None else 1, "data": "dry-run (no kernel)" if a.dry_run else "synthetic", "iters": a.iters, "rollout_s_per_iter": t_roll / a.iters, "update_s_per_iter": t_upd / a.iters,
            "rollout_env_steps_per_s": samples / t_roll, "minibatches_per_iter": a.epochs * -(-a.n_steps * a.envs // a.batch),
            "config": {"workload": f"{a.vehicle} PPO MLP[128,64,64] tanh", "envs_per_gpu": a.envs, "global_envs": a.envs * sh.world,
                       "parallelism": f"env-shard x{sh.world}; one all-reduce of the flat gradient per minibatch", "fused_rollout": bool(a.fused_rollout), "n_steps": a.n_steps, "batch_size": a.batch,
                       "n_epochs": a.epochs, "parameters": algo.policy.num_parameters(), "grad_allreduce_bytes": 4 * algo.policy.num_parameters()},
            "dtype": "f32", "last_losses": rec}))
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
