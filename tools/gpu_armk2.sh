#!/bin/bash
# whole GPU suite + crossover of the stage-wave kernel against the two-wave kernel.   bash tools/gpu_armk2.sh
set -o pipefail
O=gpurun_out/armk; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_all.log 2>&1; E=$?
tail -8 $O/pytest_all.log; [ $E -eq 0 ] || exit $E
for N in 6400 12288 24576 36864 40960 45056 49152 53248; do
  for K in staged helper; do
    timeout -k 10 120 python bench.py --vehicle hexa_arm --envs-per-gpu $N --kernel $K --steps 512 --warmup 64 --repeats 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$N $K', 'us/step %.2f' % (d['ms_per_step'] * 1e3), 'kernel_us %.2f' % d['roofline'].get('kernel_us', 0), d['config'].get('kernel', '')[:40])" || exit 1
  done
done | tee $O/sweep2.txt
