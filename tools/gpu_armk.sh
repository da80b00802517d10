#!/bin/bash
# stage-wave arm kernel: parity tests, then launch time against the other kernels by batch size.   bash tools/gpu_armk.sh
set -o pipefail
O=gpurun_out/armk; mkdir -p $O
[ -n "$SKIP_TESTS" ] || timeout -k 10 500 python -m pytest tests/test_gpu_arm.py -x -q -p no:cacheprovider -k "closed_loop_vs_oracle or forward_kinematics or tracks_the_lane" > $O/pytest.log 2>&1; E=$?
[ -n "$SKIP_TESTS" ] || { tail -15 $O/pytest.log; [ $E -eq 0 ] || exit $E; }
for N in 4096 8192 16384 32768 65536 131072 262144; do
  for K in staged helper lane; do
    timeout -k 10 120 python bench.py --vehicle hexa_arm --envs-per-gpu $N --kernel $K --steps 512 --warmup 64 --repeats 3 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$N $K', 'us/step %.2f' % (d['ms_per_step'] * 1e3), 'kernel_us %.2f' % d['roofline'].get('kernel_us', 0), d['config'].get('kernel', '')[:40])" || exit 1
  done
done | tee $O/sweep.txt
