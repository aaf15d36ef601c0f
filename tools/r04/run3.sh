#!/bin/bash
# round 4, GPU call 3: full select with the sliced LDS hash build, element-parallel ordering, nibble treeify check
set -o pipefail
mkdir -p gpurun_out/r04_3
timeout -k 10 900 python -m pytest tests/test_gpu_shipped_profiles.py tests/test_gpu_treeify.py tests/test_gpu_route_edges.py tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu > gpurun_out/r04_3/tests.log 2>&1
rc=$?
tail -5 gpurun_out/r04_3/tests.log
[ $rc -ne 0 ] && exit $rc
for w in sift1m_P4_FAST sift1m_P10_HIGH; do
  for v in "new:" "old:FSPANN_ROUTE_SLICE=0"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs timeout -k 10 400 python bench.py --workload $w --k 100 --data clustered --steps 40 --warmup 3 --prewarm 10 --no-extras --no-cpu-baseline --no-shipped > gpurun_out/r04_3/${w}_${name}.json 2> gpurun_out/r04_3/${w}_${name}.err || { tail -5 gpurun_out/r04_3/${w}_${name}.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/r04_3/${w}_${name}.json").read().strip().splitlines()[-1])
print("${w} ${name}", d["value"], d["ms_per_step"], d["stages_ms"], d["treeified"]["flagged_and_left_empty_in_all_timed_and_untimed_runs"])
PY
  done
done
