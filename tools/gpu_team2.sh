#!/bin/bash
# team kernels after a change: the arm tests, then the default bench line (4096 envs) without the CPU baseline.   bash tools/gpu_team2.sh
set -o pipefail
O=gpurun_out/team2; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_arm.py tests/test_gpu_parity.py -x -q -p no:cacheprovider > $O/pytest.log 2>&1; E=$?
tail -6 $O/pytest.log; [ $E -eq 0 ] || exit $E
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err && python - <<'PY'
import json
d = json.loads(open("gpurun_out/team2/bench.json").readline())
print("value %.4g" % d["value"], "us/step %.3f" % (d["ms_per_step"] * 1e3), "kernel_us", d["roofline"].get("kernel_us"), d["config"]["kernel"][:50])
for k, v in d.get("extras", {}).items():
    print(k, json.dumps(v)[:200])
PY
