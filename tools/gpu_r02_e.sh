#!/bin/bash
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02e; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
python tools/pmc_traffic.py --vehicle hexa_arm --envs 4096 --out $O > $O/pmc_traffic_arm.log 2>&1; tail -2 $O/pmc_traffic_arm.log
python tools/pmc_traffic.py --vehicle hexa --envs 4096 --out $O > $O/pmc_traffic_hexa.log 2>&1; tail -2 $O/pmc_traffic_hexa.log
bash tools/pmc_sq.sh hexa_arm team > /dev/null 2>&1; cp gpurun_out/pmc_sq_hexa_arm_team.txt $O/
for N in 262144 1048576; do for K in lane helper; do
  python bench.py --kernel $K --envs-per-gpu $N --no-cpu-baseline --no-extras --steps 256 --preroll 256 > $O/b_${K}_$N.json 2>/dev/null; python -c "
import json
d=json.load(open('$O/b_${K}_$N.json')); print('$N $K dev us/step %.3f  %.4g env-steps/s  frac %.3f' % (d['device_ms_per_step']*1e3, d['value'], d['roofline']['frac']))"
done; done
