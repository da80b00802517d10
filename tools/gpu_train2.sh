#!/bin/bash
# does PPO with the reference's hyper-parameters learn the rigid hexacopter?  (the arm vehicle did not: tools/gpu_train.sh)
O=gpurun_out/train; mkdir -p $O
( time timeout -k 10 500 python examples/rl_train_gpu.py --vehicle hexa --envs 256 --n-steps 512 --timesteps 150000000 --save $O/hexa_model ) > $O/hexa.log 2>&1; tail -3 $O/hexa.log | cut -c1-300
grep -c success_rate $O/hexa.log
