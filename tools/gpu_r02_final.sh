#!/bin/bash
# round-2 evidence pass: everything profiles/r02/ cites, from one box.   bash tools/gpu_r02_final.sh   (diagnostic libs: tools/build_variant.py)
set -o pipefail
R=$PWD; O=$R/gpurun_out/r02final; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench done"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_driver -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_driver.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/prof_default.log 2>&1)
cp $O/prof_driver/*/*kernel_stats.csv $O/bench_driver_kernel_stats.csv; cp $O/prof_default/*/*kernel_stats.csv $O/bench_default_kernel_stats.csv
python tools/pmc_traffic.py --vehicle hexa_arm --envs 4096 --out $O > $O/pmc_traffic_arm_team_4096.log 2>&1
python tools/pmc_traffic.py --vehicle hexa --envs 4096 --out $O > $O/pmc_traffic_hexa.log 2>&1
echo "traffic done"
bash tools/pmc_sq.sh hexa_arm team > /dev/null 2>&1; cp gpurun_out/pmc_sq_hexa_arm_team.txt $O/pmc_sq_step_kernel_team_4096.txt
bash tools/pmc_sq.sh hexa_arm helper > /dev/null 2>&1; cp gpurun_out/pmc_sq_hexa_arm_helper.txt $O/pmc_sq_step_kernel_arm2w_4096.txt
bash tools/pmc_sq.sh hexa auto > /dev/null 2>&1; cp gpurun_out/pmc_sq_hexa_auto.txt $O/pmc_sq_step_kernel_pw_4096.txt
bash tools/pmc_sq.sh hexa_arm auto 1048576 > /dev/null 2>&1; cp gpurun_out/pmc_sq_hexa_arm_auto_1048576.txt $O/pmc_sq_step_kernel_arm_1048576.txt
bash tools/pmc_sq.sh hexa_arm staged 32768 > /dev/null 2>&1; cp gpurun_out/pmc_sq_hexa_arm_staged_32768.txt $O/pmc_sq_step_kernel_armk_32768.txt
echo "sq done"
AMENV_LIB=$R/tools/micro/libamenv_stamps.so python tools/stamp_team.py > $O/stamps_team_4096.txt 2>/dev/null
AMENV_LIB=$R/tools/micro/libamenv_stamps.so python tools/stamp_pw.py --vehicle hexa > $O/stamps_pw_hexa_4096.txt 2>/dev/null
AMENV_LIB=$R/tools/micro/libamenv_stamps.so python tools/launch_floor.py --envs 4096 > $O/launch_floor.txt 2>/dev/null
for V in hexa quad hexa_arm; do python tools/reset_cost.py --vehicle $V 2>/dev/null; done > $O/reset_cost.txt
(bash tools/gpu_cross.sh; CROSS_SIZES="4096 4608 5120 6144" bash tools/gpu_cross.sh) > $O/crossover_team_vs_arm2w.txt 2>&1
SKIP_TESTS=1 bash tools/gpu_armk.sh > /dev/null 2>&1; cat gpurun_out/armk/sweep.txt > $O/crossover_stage_wave_kernel.txt
echo "stamps done"
bash tools/gpu_r02_g.sh > $O/sweep.txt 2>&1
cat gpurun_out/r02g/vecenv_quad.json gpurun_out/r02g/vecenv_arm.json > $O/vecenv_rate.json
SKIP_TESTS=1 bash tools/gpu_ppo.sh > $O/ppo_loop.txt 2>&1; cp gpurun_out/ppo/ppo_quad.json $O/ppo_bench_quad.json; cp gpurun_out/ppo/ppo_hexa_arm.json $O/ppo_bench_arm.json
cp gpurun_out/ppo/ppo_hexa_arm_fused.json $O/ppo_bench_arm_fused_rollout.json; cp gpurun_out/ppo/ppo_update_kernel_stats.csv $O/ppo_update_kernel_stats.csv
python tools/ppo_mlp_rate.py > $O/ppo_mlp_step.txt 2>/dev/null
AMENV_LIB=$R/tools/micro/libamenv_mlpstamps.so python tools/ppo_mlp_stamps.py >> $O/ppo_mlp_step.txt 2>/dev/null
tools/micro/mfma_rate > $O/mfma_rate.txt 2>&1
echo "ppo done"
bash tools/gpu_team.sh > $O/closed_loop_policy_rollout.txt 2>&1
timeout -k 10 300 python tools/micro/f64_arm_unrolled_repro.py > $O/f64_arm_unrolled_repro.log 2>&1
ls $O | head -80
