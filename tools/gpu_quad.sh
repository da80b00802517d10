#!/bin/bash
# lane-quad rigid kernel: whole GPU suite, then launch time against the helper-wave kernel by batch size.   bash tools/gpu_quad.sh
set -o pipefail
O=gpurun_out/quad; mkdir -p $O
[ -n "$SKIP_TESTS" ] || { timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_all.log 2>&1; E=$?; tail -12 $O/pytest_all.log; [ $E -eq 0 ] || exit $E; }
for V in hexa quad; do
  for N in 1024 4096 8192 16384 24576 32768; do
    for K in team helper; do
      timeout -k 10 120 python bench.py --vehicle $V --envs-per-gpu $N --kernel $K --steps 1024 --warmup 64 --repeats 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$V $N $K', 'us/step %.3f' % (d['ms_per_step'] * 1e3), 'kernel_us %.3f' % d['roofline'].get('kernel_us', 0), d['config'].get('kernel', '')[:36])" || exit 1
    done
  done
done | tee $O/sweep.txt
