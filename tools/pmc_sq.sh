# SQ counters of the step kernel (separate --pmc passes), summarised by tools/pmc_summary.py
# usage: bash tools/pmc_sq.sh <vehicle> [kernel] [envs]   -> gpurun_out/pmc_sq_<vehicle>_<kernel>[_<envs>].txt
V=${1:-hexa_arm}; R=$PWD; export TMPDIR=/tmp; N=${3:-4096}; S=$([ $N -gt 100000 ] && echo 30 || echo 120)
T=${V}_${2:-auto}$([ $N != 4096 ] && echo _$N)
: > $R/gpurun_out/pmc_sq_$T.txt
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM"; do
  D=$R/gpurun_out/pmc_sq_${T}_$(echo $C | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $R/tools/pmc_step.py --envs $N --steps $S --vehicle $V --kernel ${2:-auto} > /dev/null 2>&1)
  python3 $R/tools/pmc_summary.py $D step_kernel >> $R/gpurun_out/pmc_sq_$T.txt
done
cat $R/gpurun_out/pmc_sq_$T.txt
