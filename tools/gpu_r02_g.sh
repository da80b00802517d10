#!/bin/bash
R=$PWD; O=$R/gpurun_out/r02g; mkdir -p $O
python tools/vecenv_rate.py --vehicle quad > $O/vecenv_quad.json 2>/dev/null; cat $O/vecenv_quad.json
python tools/vecenv_rate.py --vehicle hexa_arm > $O/vecenv_arm.json 2>/dev/null; cat $O/vecenv_arm.json
for V in hexa quad; do python bench.py --vehicle $V --no-cpu-baseline --no-extras > $O/bench_$V.json 2>/dev/null; python -c "
import json
d=json.load(open('$O/bench_$V.json')); print('$V dev us/step %.3f  %.4g env-steps/s  %s' % (d['device_ms_per_step']*1e3, d['value'], d['config']['kernel'][:40]))"; done
for N in 32768 262144 1048576; do python bench.py --envs-per-gpu $N --steps 512 --preroll 512 --no-cpu-baseline --no-extras > $O/bench_arm_$N.json 2>/dev/null; python -c "
import json
d=json.load(open('$O/bench_arm_$N.json')); print('hexa_arm $N dev us/step %.3f  %.4g env-steps/s frac %.3f %s' % (d['device_ms_per_step']*1e3, d['value'], d['roofline']['frac'], d['config']['kernel'][:30]))"; done
for N in 32768 1048576; do python bench.py --vehicle hexa --envs-per-gpu $N --steps 512 --preroll 512 --no-cpu-baseline --no-extras > $O/bench_hexa_$N.json 2>/dev/null; python -c "
import json
d=json.load(open('$O/bench_hexa_$N.json')); print('hexa $N dev us/step %.3f  %.4g env-steps/s frac %.3f' % (d['device_ms_per_step']*1e3, d['value'], d['roofline']['frac']))"; done
