#!/bin/bash
# round-2 GPU pass A: parity tests, the driver's bench command, kernel-trace profile of the same command, PMC traffic
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02d; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err && tail -c 600 $O/bench_driver.json &&
python bench.py > $O/bench_default.json 2> $O/bench_default.err && tail -c 300 $O/bench_default.json &&
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_driver -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_driver.log 2>&1) &&
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/prof_default.log 2>&1) &&
python tools/pmc_traffic.py --vehicle hexa_arm --envs 4096 --out $O > $O/pmc_traffic_arm.log 2>&1; tail -3 $O/pmc_traffic_arm.log
python tools/pmc_traffic.py --vehicle hexa --envs 4096 --out $O > $O/pmc_traffic_hexa.log 2>&1; tail -3 $O/pmc_traffic_hexa.log
ls $O
