#!/bin/bash
# shipped profiles on SIFT-like and clustered data after the full select's repeat / group / cap changes
set -o pipefail
mkdir -p gpurun_out/r04_12
for w in sift1m_P4_FAST sift1m_P10_HIGH; do
  for dk in "siftlike:16:6" "clustered"; do
    name=${dk//:/_}
    timeout -k 10 300 python bench.py --workload $w --k 100 --data $dk --steps 40 --warmup 3 --prewarm 10 --no-extras --no-shipped --cpu-sample 64 --solo-tail 0 > gpurun_out/r04_12/${w}_${name}.json 2> gpurun_out/r04_12/${w}_${name}.err || { tail -5 gpurun_out/r04_12/${w}_${name}.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/r04_12/${w}_${name}.json").read().strip().splitlines()[-1])
print("${w} ${dk}", d["value"], d["ms_per_step"], d["stages_ms"], "recall", d.get("recall_at_k"), "frac", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("matches_gpu"))
PY
  done
done
