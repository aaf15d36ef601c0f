#!/bin/bash
# round 4, GPU call 1: the bounded select's exact treeify check — tests, then what the check costs at config #2
set -o pipefail
mkdir -p gpurun_out/r04_1
timeout -k 10 900 python -m pytest tests/test_gpu_treeify.py tests/test_gpu_route_edges.py tests/test_gpu_parity.py tests/test_gpu_tick.py -x -q -m gpu > gpurun_out/r04_1/tests.log 2>&1
rc=$?
tail -5 gpurun_out/r04_1/tests.log
[ $rc -ne 0 ] && exit $rc
for v in "default:" "small0:FSPANN_ROUTE_LAZY_SMALL=0" "check1:FSPANN_ROUTE_BINCHECK=1"; do
  name=${v%%:*}; envs=${v#*:}
  for pipe in front serial; do
    env $envs timeout -k 10 300 python bench.py --steps 400 --warmup 20 --no-extras --no-cpu-baseline --pipeline $pipe > gpurun_out/r04_1/bench_${name}_${pipe}.json 2> gpurun_out/r04_1/bench_${name}_${pipe}.err || { tail -5 gpurun_out/r04_1/bench_${name}_${pipe}.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/r04_1/bench_${name}_${pipe}.json").read().strip().splitlines()[-1])
print("${name} ${pipe}", d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"))
PY
  done
done
