#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04_spec
timeout -k 10 900 python -m pytest tests/test_gpu_route_edges.py tests/test_gpu_parity.py tests/test_gpu_tick.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for v in "spec:" "gen:FSPANN_ROUTE_SHAPE_SPEC=0" "spec2:" "gen2:FSPANN_ROUTE_SHAPE_SPEC=0"; do
  name=${v%%:*}; envs=${v#*:}
  for pipe in front serial; do
    env $envs timeout -k 10 300 python bench.py --steps 400 --warmup 20 --no-extras --no-cpu-baseline --no-shipped --pipeline $pipe > gpurun_out/r04_spec/bench_${name}_${pipe}.json 2> gpurun_out/r04_spec/bench_${name}_${pipe}.err || { tail -5 gpurun_out/r04_spec/bench_${name}_${pipe}.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/r04_spec/bench_${name}_${pipe}.json").read().strip().splitlines()[-1])
print("${name} ${pipe}", d["value"], d["ms_per_step"], d["stages_ms"]["route_select"])
PY
  done
done
