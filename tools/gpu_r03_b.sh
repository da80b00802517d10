#!/bin/bash
# round 3, quick pass: team-kernel timing only (bench default without extras, rocprof kernel stats of the driver command, SQ VALU count, stamps)
set -o pipefail
R=$PWD; O=$R/gpurun_out/${1:-r03b}; mkdir -p $O; export TMPDIR=/tmp
python rl-aerial-manipulator_amd/build.py > /dev/null 2>&1
python -m pytest tests/test_gpu_arm.py -x -q -p no:cacheprovider -k "closed_loop or team" > $O/pytest_team.log 2>&1; echo "pytest exit $?"; tail -2 $O/pytest_team.log
python bench.py --no-extras --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; python - <<PY
import json; d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1]); print("default: us/step", d["ms_per_step"]*1e3, "value", d["value"], "roofline", d.get("roofline"))
PY
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_driver -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_driver.log 2>&1)
cp $O/prof_driver/*/*kernel_stats.csv $O/bench_driver_kernel_stats.csv; head -2 $O/bench_driver_kernel_stats.csv | cut -c1-60,200-300
(cd /tmp && timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/pmc -- python3 $R/tools/pmc_step.py --envs 4096 --steps 120 --vehicle hexa_arm --kernel team > /dev/null 2>&1); python3 tools/pmc_summary.py $O/pmc step_kernel | tee $O/pmc_valu.txt
if [ -f tools/micro/libamenv_stamps.so ]; then AMENV_LIB=$R/tools/micro/libamenv_stamps.so python tools/stamp_team.py --launches 600 > $O/stamps_team_4096.txt 2>$O/stamps.err; cat $O/stamps_team_4096.txt; fi
