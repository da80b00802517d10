#!/bin/bash
R=$PWD; O=$R/gpurun_out/cross; mkdir -p $O
for N in ${CROSS_SIZES:-2048 8192 12288 16384 24576}; do for K in team helper; do
  python bench.py --kernel $K --envs-per-gpu $N --no-cpu-baseline --no-extras --steps 2048 > $O/b_${K}_$N.json 2>/dev/null; python -c "
import json
d=json.load(open('$O/b_${K}_$N.json')); print('$N $K dev us/step %.3f  %.4g env-steps/s' % (d['device_ms_per_step']*1e3, d['value']))"
done; done
