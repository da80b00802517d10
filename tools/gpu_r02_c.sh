#!/bin/bash
# round-2 GPU pass C: team kernel correctness + A/B timing against the two-wave kernel
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02c; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_gpu_arm.py -m gpu -x -q -p no:cacheprovider > $O/pytest_arm.log 2>&1; echo "pytest exit $?" | tee -a $O/pytest_arm.log
tail -25 $O/pytest_arm.log
for K in helper team lane; do
  python bench.py --kernel $K --no-cpu-baseline --no-extras > $O/bench_$K.json 2> $O/bench_$K.err; python -c "
import json,sys
d=json.load(open('$O/bench_$K.json')); print('$K', d['value'], d['device_ms_per_step'], d['roofline']['kernel_us_isolated_min'], d['config']['kernel'])"
done
for K in helper team; do
  python bench.py --kernel $K --envs-per-gpu 32768 --no-cpu-baseline --no-extras > $O/bench_${K}_32768.json 2> $O/bench_${K}_32768.err; python -c "
import json,sys
d=json.load(open('$O/bench_${K}_32768.json')); print('$K 32768', d['value'], d['device_ms_per_step'])"
done
