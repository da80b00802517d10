#!/bin/bash
# round-2 GPU pass B: parity tests after the HBM re-layout and the fp64 arm kernel; record of the unrolled (spilling) fp64 build
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02b; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log
timeout -k 10 300 python tools/micro/f64_arm_unrolled_repro.py > $O/f64_arm_unrolled_repro.log 2>&1; cat $O/f64_arm_unrolled_repro.log | tail -5
