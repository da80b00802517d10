#!/bin/bash
# A/B of library builds on the headline workload: bash tools/gpu_r03_ab.sh <out> libA.so libB.so ...
set -o pipefail
R=$PWD; O=$R/gpurun_out/$1; shift; mkdir -p $O; export TMPDIR=/tmp
python tools/ab_kernel.py --vehicle hexa_arm --envs 4096 --rounds 3 --libs "$@" 2>&1 | tee $O/ab.txt
for L in "$@"; do case $L in *stamps*) echo "== $L"; AMENV_LIB=$R/$L python tools/stamp_team.py --launches 400 2>/dev/null | tee -a $O/stamps.txt;; esac; done
