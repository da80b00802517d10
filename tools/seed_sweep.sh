#!/bin/bash
# PPO from scratch on the reference's quadrotor task, several seeds: how often does the reference's recipe leave the hover plateau?
#   bash tools/seed_sweep.sh "1 2 3 4" [extra rl_train_gpu.py args]   -> gpurun_out/seeds/seed_<s>.json, one summary line per seed
O=gpurun_out/seeds; mkdir -p $O
for S in $1; do
  timeout -k 10 200 python examples/rl_train_gpu.py --envs 256 --n-steps 512 --timesteps 90000000 --fused-rollout --seed $S ${@:2} --save /tmp/wp_seed_$S --log-json $O/seed_$S.json > $O/seed_$S.log 2>&1
  python -c "
import json; d=json.load(open('$O/seed_$S.json')); print('seed $S', round(d['learn_seconds'],1), 's  >90% from', d['timesteps_to_90pct_success'], ' final success', round(d['final_success_rate_mean_of_last_10_iterations'],4), ' eval', round(d['evaluate_policy_mean_reward']))"
done
