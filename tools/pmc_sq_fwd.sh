# SQ counters of the forward-only MLP kernel at 1 M rows (separate --pmc passes).   bash tools/pmc_sq_fwd.sh -> gpurun_out/pmc_sq_mlp_forward_kernel.txt
R=$PWD; export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_sq_mlp_forward_kernel.txt; : > $OUT
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_MISC"; do
  D=$R/gpurun_out/pmc_sq_fwd_$(echo $C | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $R/tools/ppo_mlp_rate.py > /dev/null 2>&1)
  python3 $R/tools/pmc_summary.py $D mlp_forward_kernel >> $OUT
done
cat $OUT
