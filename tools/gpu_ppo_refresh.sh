#!/bin/bash
# PPO evidence after a change of the update kernels: tests, loop numbers, kernel stats, forward+backward rate.   bash tools/gpu_ppo_refresh.sh
set -o pipefail
O=gpurun_out/ppo; mkdir -p $O
bash tools/gpu_ppo.sh > $O/ppo_loop.txt 2>&1; E=$?; tail -14 $O/ppo_loop.txt | cut -c1-200; [ $E -eq 0 ] || exit $E
python tools/ppo_mlp_rate.py > $O/ppo_mlp_step.txt 2>/dev/null && AMENV_LIB=$PWD/tools/micro/libamenv_mlpstamps.so python tools/ppo_mlp_stamps.py >> $O/ppo_mlp_step.txt 2>/dev/null; cat $O/ppo_mlp_step.txt
