set -o pipefail
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ppo.py -q -k "other_rigid or evaluate" > gpurun_out/new_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/new_tests.log; tail -5 gpurun_out/new_tests.log
run() { echo "== $*"; timeout -k 10 120 env "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,3),'e9', round(d['ms_per_step']*1e3,2),'us', d['config']['kernel'])"; }
for bs in 64 128 256; do run X=1 python bench.py --no-cpu-baseline --vehicle hexa --envs-per-gpu 32768 --block-size $bs --steps 8192 --warmup 512; done
for bs in 64 256; do run X=1 python bench.py --no-cpu-baseline --vehicle hexa --envs-per-gpu 262144 --block-size $bs --steps 2048 --warmup 128; done
run AMENV_ARM_2WAVE=1 python bench.py --no-cpu-baseline --envs-per-gpu 65536 --steps 8192 --warmup 512
run AMENV_ARM_2WAVE=0 python bench.py --no-cpu-baseline --envs-per-gpu 65536 --steps 8192 --warmup 512
for bs in 64 128 256; do run AMENV_ARM_2WAVE=0 python bench.py --no-cpu-baseline --envs-per-gpu 32768 --block-size $bs --steps 8192 --warmup 512; done
