# SQ counters of the fused PPO minibatch kernel (separate --pmc passes), summarised by tools/pmc_summary.py
# usage: bash tools/pmc_sq_mlp.sh   -> gpurun_out/pmc_sq_ppo_mlp_fused_kernel.txt
R=$PWD; export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_sq_ppo_mlp_fused_kernel.txt; : > $OUT
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  D=$R/gpurun_out/pmc_sq_mlp_$(echo $C | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $R/tools/ppo_mlp_stamps.py > /dev/null 2>&1)
  python3 $R/tools/pmc_summary.py $D ppo_mlp_fused_kernel >> $OUT
done
cat $OUT
