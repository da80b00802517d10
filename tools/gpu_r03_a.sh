#!/bin/bash
# round 3, pass A: the rewritten lane-team kernel -- arm tests, bench (driver + default), rocprof kernel stats, SQ counters, stamps
set -o pipefail
R=$PWD; O=$R/gpurun_out/r03a; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_arm.py -x -q -p no:cacheprovider > $O/pytest_arm.log 2>&1; echo "pytest arm exit $?" | tee -a $O/pytest_arm.log; tail -5 $O/pytest_arm.log
python bench.py --steps 20 --warmup 5 --no-extras > $O/bench_driver.json 2> $O/bench_driver.err; cat $O/bench_driver.json | cut -c1-600
python bench.py --no-extras --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json | cut -c1-400
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_driver -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_driver.log 2>&1)
cp $O/prof_driver/*/*kernel_stats.csv $O/bench_driver_kernel_stats.csv; head -3 $O/bench_driver_kernel_stats.csv
bash tools/pmc_sq.sh hexa_arm team > /dev/null 2>&1; cp gpurun_out/pmc_sq_hexa_arm_team.txt $O/pmc_sq_step_kernel_team_4096.txt; cat $O/pmc_sq_step_kernel_team_4096.txt
AMENV_LIB=$R/tools/micro/libamenv_stamps.so python tools/stamp_team.py > $O/stamps_team_4096.txt 2>$O/stamps.err; cat $O/stamps_team_4096.txt
python tools/reset_cost.py --vehicle hexa_arm 2>/dev/null | tee $O/reset_cost.txt
