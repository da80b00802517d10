#!/bin/bash
# quick validation pass: GPU tests, smoke, the driver's bench form and the default bench.   bash tools/gpu_check.sh
set -o pipefail
R=$PWD; O=$R/gpurun_out/check; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=15 > $O/pytest_gpu.log 2>&1; E=$?; echo "pytest exit $E" | tee -a $O/pytest_gpu.log; tail -25 $O/pytest_gpu.log
[ $E -eq 0 ] || exit $E
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && tail -1 $O/smoke.log &&
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err && cat $O/bench_driver.json &&
python bench.py > $O/bench_default.json 2> $O/bench_default.err && cat $O/bench_default.json
