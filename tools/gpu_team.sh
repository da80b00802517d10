#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/team; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_gpu_arm.py -m gpu -x -q -p no:cacheprovider -k "policy_rollout" > $O/pytest_team.log 2>&1; echo "pytest exit $?"; grep -E "^E|passed|failed" $O/pytest_team.log | head -20
for OCC in auto; do python - <<'PY'
import torch, os
import rl_aerial_manipulator_amd as amd
for n in (4096, 8192, 32768):
    T = 64
    pol = amd.ActorCritic(29, 7).cuda().flatten_()
    env = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=0)
    env.reset()
    dev = env.device
    obs = torch.zeros(T + 1, n, 29, device=dev); acts = torch.zeros(T, n, 7, device=dev)
    logp = torch.zeros(T, n, device=dev); vals = torch.zeros(T, n, device=dev); rew = torch.zeros(T, n, device=dev)
    dones = torch.zeros(T, n, dtype=torch.uint8, device=dev)
    run = lambda: env.rollout_policy(pol.flat_param, T, seed=1, draw0=0, obs=obs, actions=acts, logp=logp, values=vals, rewards=rew, dones=dones)
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (8 * T)
    print(f"closed-loop policy rollout n={n}: {us:.3f} us/step  {n / us * 1e6:.4g} env-steps/s")
PY
done
