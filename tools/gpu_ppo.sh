#!/bin/bash
# PPO loop numbers (config 5) + kernel stats of the update.   bash tools/gpu_ppo.sh
R=$PWD; O=$R/gpurun_out/ppo; mkdir -p $O; export TMPDIR=/tmp
[ -n "$SKIP_TESTS" ] || { timeout -k 10 500 python -m pytest tests/test_gpu_ppo.py -x -q 2>&1 | tail -5 || exit 1; }
for V in quad hexa_arm; do timeout -k 10 200 python tools/ppo_bench.py --vehicle $V > $O/ppo_$V.json 2>$O/ppo_$V.err || exit 1; python -c "import json; d=json.load(open('$O/ppo_$V.json')); print('$V', {k:d[k] for k in ('value','rollout_s_per_iter','update_s_per_iter')})"; done
timeout -k 10 200 python tools/ppo_bench.py --vehicle hexa_arm --fused-rollout > $O/ppo_hexa_arm_fused.json 2>$O/ppo_fused.err || exit 1
python -c "import json; d=json.load(open('$O/ppo_hexa_arm_fused.json')); print('fused', {k:d[k] for k in ('value','rollout_s_per_iter','update_s_per_iter')})"
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/ppo_bench.py --vehicle hexa_arm --fused-rollout --iters 1 --warmup 1 > /dev/null 2>&1)
F=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $F $O/ppo_update_kernel_stats.csv; head -14 $F | cut -c1-170
