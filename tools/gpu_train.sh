#!/bin/bash
# PPO from scratch on the HIP env with the reference's hyper-parameters: the quadrotor run of profiles/r01 again (time to solution with the fused
# update) and the hexacopter + arm (tool-point task).   bash tools/gpu_train.sh
O=gpurun_out/train; mkdir -p $O
( time timeout -k 10 420 python examples/rl_train_gpu.py --envs 256 --n-steps 512 --timesteps 90000000 --save $O/quad_model ) > $O/quad.log 2>&1; tail -4 $O/quad.log | cut -c1-300
( time timeout -k 10 600 python examples/rl_train_gpu.py --vehicle hexa_arm --envs 256 --n-steps 512 --timesteps 150000000 --save $O/arm_model ) > $O/arm.log 2>&1; tail -4 $O/arm.log | cut -c1-300
