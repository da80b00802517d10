#!/bin/bash
# lane-team (stage-parallel RK4) against the stage-wave kernel around the AUTO limit.   bash tools/gpu_cross2.sh
for N in 4096 5120 6144 7168 8192 10240; do
  for K in team staged; do
    timeout -k 10 120 python bench.py --vehicle hexa_arm --envs-per-gpu $N --kernel $K --steps 1024 --warmup 64 --repeats 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$N $K', 'us/step %.3f' % (d['ms_per_step'] * 1e3), d['config'].get('kernel', '')[:28])" || exit 1
  done
done
