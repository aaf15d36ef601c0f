#!/bin/bash
# fixed cost of the timed region: ms_per_step at 20 / 40 / 100 / 400 steps, and with the runtime's active-wait knob
set -o pipefail
mkdir -p gpurun_out/r04_fill
for v in "plain:" "spin:ROC_ACTIVE_WAIT_TIMEOUT=2000" "plain2:" "spin2:ROC_ACTIVE_WAIT_TIMEOUT=2000"; do
  name=${v%%:*}; envs=${v#*:}
  for st in 20 40 100 400; do
    env $envs timeout -k 10 300 python bench.py --steps $st --warmup 5 --no-extras --no-cpu-baseline --no-shipped > gpurun_out/r04_fill/b_${name}_${st}.json 2> gpurun_out/r04_fill/b_${name}_${st}.err || { tail -5 gpurun_out/r04_fill/b_${name}_${st}.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/r04_fill/b_${name}_${st}.json").read().strip().splitlines()[-1])
print("${name} steps ${st}", d["value"], d["ms_per_step"], "issue", d.get("host_issue_ms_per_step"), "region us", round(d["ms_per_step"]*${st}*1000,1))
PY
  done
done
