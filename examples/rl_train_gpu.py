"""The reference's training entry (initial-implementation-v2/rl_train.py:22-64) on the GPU-resident stack.

    python examples/rl_train_gpu.py [--envs 4096] [--timesteps 4100000] [--resume ppo_model_2300000_steps.zip]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/rl_train_gpu.py ...

Same hyper-parameters as the reference (lr 2e-4, 12 epochs, gamma .995, lambda .9, clip .2, ent 5e-4 -- 1e-4 when resuming,
:35 -- MLP [128,64,64] tanh); the rollout is `envs x n_steps` instead of `8 x 2048`, so n_steps / batch_size are rescaled to
keep the reference's 16,384-sample rollouts x 128-sample minibatches ratio (128 minibatches per epoch).

Defaults = the run recorded in profiles/r01/ppo_from_scratch.json: 256 envs x 512 steps, 90 M steps, ~10 minutes on one MI355X, > 90 % of
episodes successful after 39 M steps.  Learning progress follows the number of Adam updates (1536 per iteration here), not the number of samples:
thousands of envs with proportionally larger minibatches collect samples faster but learn no sooner.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=256)
    ap.add_argument("--n-steps", type=int, default=512)
    ap.add_argument("--timesteps", type=int, default=90_000_000)         # rl_train.py:56 uses 4.1 M per run, ~15 M in total
    ap.add_argument("--resume", default=None, help="SB3 zip / policy.pth to start from (rl_train.py:33-35)")
    ap.add_argument("--save", default="waypoint_controller_gpu")         # rl_train.py:57
    ap.add_argument("--vehicle", default="quad")
    ap.add_argument("--seed", type=int, default=0, help="env reset stream, initial weights and action noise (one run = one seed: PPO on this task is seed-sensitive)")
    ap.add_argument("--moment-scale", type=float, default=None, help="N m per unit moment action (amenv_vehicle.moment_scale; the reference quadrotor: 0.1)")
    ap.add_argument("--fused-rollout", action="store_true", help="collect every rollout as ONE launch (amenv_rollout_policy: the policy on bf16 matrix cores inside the env loop; "
                                                                 "log-probs / values of the buffer re-evaluated in fp32); quadrotor, hexacopter, hexacopter + arm")
    ap.add_argument("--log-json", default=None, help="write the learning curve (one record per iteration) and the final evaluation to this file")
    ap.add_argument("--warm-start-pid", type=int, default=None, metavar="DAGGER_ROUNDS",
                    help="initialise the actor by behaviour cloning of the PID + minimum-snap baseline (amd.clone_pid_policy; 0 = plain cloning, k = k DAgger rounds). "
                         "Needed for the hexacopter vehicles: from SB3's default initialisation PPO does not leave the free-fall plateau there (profiles/r03/ppo_hexa_sweep_*.json)")
    a = ap.parse_args()
    import torch
    import rl_aerial_manipulator_amd as amd
    from rl_aerial_manipulator_amd import sharding
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    sh = sharding.shard_from_env(a.envs)
    dist = sharding.init_process_group("nccl", torch.device("cuda", local))
    cfg = None
    if a.moment_scale is not None:
        cfg = amd._lib.default_config(a.vehicle, a.envs)
        cfg.vehicle.moment_scale = a.moment_scale
        cfg.seed, cfg.env_id_offset = a.seed, sh.env_id_offset
    env = amd.GpuWaypointEnv(a.envs, device=local, vehicle=a.vehicle, seed=a.seed, env_id_offset=sh.env_id_offset, config=cfg)
    model = amd.PPO(env, learning_rate=2e-4, n_steps=a.n_steps, batch_size=a.envs * a.n_steps // 128, n_epochs=12, gamma=0.995,
                    gae_lambda=0.9, clip_range=0.2, ent_coef=1e-4 if a.resume else 5e-4, dist=dist, fused_rollout=a.fused_rollout, seed=a.seed)
    if a.resume:
        model.load_policy(a.resume)
    elif a.warm_start_pid is not None:
        mse = amd.clone_pid_policy(env, model.policy, dagger_rounds=a.warm_start_pid)
        if dist is not None:        # every rank cloned on its own shard: continue from rank 0's actor
            for p_ in model.policy.parameters():
                dist.broadcast(p_.data, src=0)
        if sh.rank == 0:
            print(f"warm start: actor cloned from PidWaypointPolicy (mse {mse:.4f}, {a.warm_start_pid} DAgger rounds), log_std = -1", flush=True)
    curve = []

    def show(r):
        curve.append({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
        print(curve[-1], flush=True)

    import time
    t0 = time.time()
    model.learn(a.timesteps, log_fn=show if sh.rank == 0 else None)
    seconds = time.time() - t0
    if sh.rank == 0:
        print("saved", model.save(a.save))
        mean_reward, std_reward = amd.evaluate_policy(model, env, n_eval_episodes=max(10, a.envs))      # rl_train.py:60-61
        print(f"Mean reward: {mean_reward} +/- {std_reward}")
        if curve and curve[-1].get("success_rate", 1.0) < 0.5 and a.warm_start_pid is None and not a.resume:
            print("this run stayed on the hover plateau (about one seed in six does within 90 M steps, profiles/r03/ppo_seed_sweep_quad/): "
                  "try another --seed, or --warm-start-pid 3 (actor cloned from the PID baseline), which leaves it from the first iterations")
        if a.log_json:
            import json
            first90 = next((c["timesteps"] for c in curve if c["success_rate"] > 0.9), None)
            with open(a.log_json, "w") as f:
                json.dump({"command": " ".join(sys.argv), "learn_seconds": seconds, "timesteps": a.timesteps, "timesteps_to_90pct_success": first90,
                           "final_success_rate_mean_of_last_10_iterations": sum(c["success_rate"] for c in curve[-10:]) / max(1, len(curve[-10:])),
                           "evaluate_policy_mean_reward": mean_reward, "evaluate_policy_std_reward": std_reward, "curve_every_20th_iteration": curve[::20]}, f, indent=1)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
