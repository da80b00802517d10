#!/bin/bash
# round-3 evidence pass: everything profiles/r03/ cites, from one box.   bash tools/gpu_r03_final.sh   (diagnostic libs: tools/build_variant.py)
set -o pipefail
R=$PWD; O=$R/gpurun_out/r03final; mkdir -p $O; export TMPDIR=/tmp
rm -f gpurun_out/threshold_flips.txt
python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $O/pytest_gpu.log; tail -3 $O/pytest_gpu.log
cp gpurun_out/threshold_flips.txt $O/threshold_flips.txt 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench done"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_driver -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/prof_driver.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --no-cpu-baseline --no-extras > $O/prof_default.log 2>&1)
cp $O/prof_driver/*/*kernel_stats.csv $O/bench_driver_kernel_stats.csv; cp $O/prof_default/*/*kernel_stats.csv $O/bench_default_kernel_stats.csv
head -2 $O/bench_driver_kernel_stats.csv | cut -c1-50,200-320; head -2 $O/bench_default_kernel_stats.csv | cut -c1-50,200-320
python tools/pmc_traffic.py --vehicle hexa_arm --envs 4096 --out $O > $O/pmc_traffic_arm_team_4096.log 2>&1; tail -3 $O/pmc_traffic_arm_team_4096.log
echo "traffic done"
bash tools/pmc_sq.sh hexa_arm team > /dev/null 2>&1; cp gpurun_out/pmc_sq_hexa_arm_team.txt $O/pmc_sq_step_kernel_team_4096.txt; head -3 $O/pmc_sq_step_kernel_team_4096.txt
bash tools/pmc_sq.sh hexa_arm staged 32768 > /dev/null 2>&1; cp gpurun_out/pmc_sq_hexa_arm_staged_32768.txt $O/pmc_sq_step_kernel_armk_32768.txt
echo "sq done"
AMENV_LIB=$R/tools/micro/libamenv_stamps.so python tools/stamp_team.py > $O/stamps_team_4096.txt 2>/dev/null; cat $O/stamps_team_4096.txt
for V in hexa_arm; do python tools/reset_cost.py --vehicle $V 2>/dev/null; done > $O/reset_cost.txt; cat $O/reset_cost.txt
bash tools/gpu_r03_ab.sh r03final_ab rl-aerial-manipulator_amd/libamenv.so tools/micro/libamenv_nork4.so tools/micro/libamenv_nohelper.so tools/micro/libamenv_norknoh.so tools/micro/libamenv_nopredict.so tools/micro/libamenv_noprio.so > $O/team_step_anatomy.txt 2>&1; cat $O/team_step_anatomy.txt
tools/micro/dpp_issue > $O/dpp_issue.txt 2>&1
bash tools/gpu_cross2.sh > $O/crossover_team_vs_stage_wave.txt 2>&1; cat $O/crossover_team_vs_stage_wave.txt
echo "anatomy done"
bash tools/gpu_r02_g.sh > $O/sweep.txt 2>&1; cat $O/sweep.txt | cut -c1-200
cat gpurun_out/r02g/vecenv_quad.json gpurun_out/r02g/vecenv_arm.json > $O/vecenv_rate.json
SKIP_TESTS=1 bash tools/gpu_ppo.sh > $O/ppo_loop.txt 2>&1; cp gpurun_out/ppo/ppo_quad.json $O/ppo_bench_quad.json; cp gpurun_out/ppo/ppo_hexa_arm.json $O/ppo_bench_arm.json
cp gpurun_out/ppo/ppo_hexa_arm_fused.json $O/ppo_bench_arm_fused_rollout.json; tail -5 $O/ppo_loop.txt | cut -c1-300
timeout -k 10 300 python tools/ppo_bench.py --vehicle quad --fused-rollout > $O/ppo_bench_quad_fused_rollout.json 2>/dev/null; cut -c1-400 $O/ppo_bench_quad_fused_rollout.json
timeout -k 10 300 python tools/micro/f64_arm_unrolled_repro.py > $O/f64_arm_unrolled_repro.log 2>&1
ls $O | head -80
