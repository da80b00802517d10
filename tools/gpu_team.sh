#!/bin/bash
# team-kernel iteration: correctness, stamp profile, timing
set -o pipefail
R=$PWD; O=$R/gpurun_out/team; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_gpu_arm.py -m gpu -x -q -p no:cacheprovider -k "team or closed_loop or forward_kin" > $O/pytest_team.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest_team.log
AMENV_LIB=$R/tools/micro/libamenv_stamps.so python tools/stamp_profile.py --envs 4096 --vehicle hexa_arm --kernel team 2>/dev/null | head -9
for K in team helper; do
  python bench.py --kernel $K --no-cpu-baseline --no-extras > $O/bench_$K.json 2> $O/bench_$K.err; python -c "
import json,sys
d=json.load(open('$O/bench_$K.json')); print('$K', '%.4g' % d['value'], 'dev us/step %.3f' % (d['device_ms_per_step']*1e3), 'isolated min %.2f' % d['roofline']['kernel_us_isolated_min'])"
done
python - <<'PY'
import torch, time
import rl_aerial_manipulator_amd as amd
for kern in ("team", "lane"):
    n, T = 4096, 64
    env = amd.GpuWaypointEnv(n, vehicle="hexa_arm", seed=0, kernel=kern)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    a = (torch.randn(T, n, 7, device="cuda", generator=g) * 0.1); a[..., 0] += 1; a[..., 4:] *= 3; a = a.clamp(-1, 2).contiguous()
    for want in (True, False):
        env.rollout(a, want_obs=want)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8): env.rollout(a, want_obs=want)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (8 * T)
        print(f"rollout {kern} obs={want}: {us:.3f} us/step  {n / us * 1e6:.4g} env-steps/s")
PY
