#!/bin/bash
# per-query calls at the shipped profiles: zero copy vs copies (long result lists written over the bus?)
set -o pipefail
mkdir -p gpurun_out/r04_17
for w in sift1m_P4_FAST sift1m_P10_HIGH; do
 for v in "zc:" "copy:FSPANN_ZERO_COPY=0"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs timeout -k 10 300 python bench.py --workload $w --k 100 --data siftlike:16:6 --steps 10 --warmup 2 --prewarm 2 --no-shipped --no-cpu-baseline --solo-tail 0 > gpurun_out/r04_17/${w}_${name}.json 2> gpurun_out/r04_17/${w}_${name}.err || { tail -5 gpurun_out/r04_17/${w}_${name}.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_17/${w}_${name}.json").read().strip().splitlines()[-1])
o=d["operator_surface"]["per_query"]
print("${w} ${name}", "enc", o["encode"], "route", o["route"], "refine", o["refine_f64_rows"], "pageable", o["refine_f64_rows_from_pageable_memory"], "r+r", o["route_plus_refine"])
PY
 done
done
