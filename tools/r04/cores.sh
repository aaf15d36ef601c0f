#!/bin/bash
# shipped profiles: can the full select and the scan share the CUs?  (Route planned with less LDS, scan as a thin grid)
set -o pipefail
mkdir -p gpurun_out/r04_cores
for w in sift1m_P10_HIGH sift1m_P4_FAST; do
  for v in "base:" "scan1:FSPANN_REFINE_STREAM=1" "scan2:FSPANN_REFINE_STREAM=2" \
           "r118_scan1:FSPANN_ROUTE_LDS_KB=118 FSPANN_REFINE_STREAM=1" "r118_base:FSPANN_ROUTE_LDS_KB=118" \
           "r80_scan2:FSPANN_ROUTE_LDS_KB=80 FSPANN_ROUTE_WGS=1 FSPANN_REFINE_STREAM=2" "r80_w2:FSPANN_ROUTE_LDS_KB=80 FSPANN_ROUTE_WGS=2" \
           "r80_w1:FSPANN_ROUTE_LDS_KB=80 FSPANN_ROUTE_WGS=1"; do
    name=${v%%:*}; envs=${v#*:}
    env $envs timeout -k 10 300 python bench.py --workload $w --k 100 --data clustered --steps 40 --warmup 3 --prewarm 10 --no-extras --no-cpu-baseline --no-shipped --solo-tail 0 > gpurun_out/r04_cores/${w}_${name}.json 2> gpurun_out/r04_cores/${w}_${name}.err || { tail -5 gpurun_out/r04_cores/${w}_${name}.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/r04_cores/${w}_${name}.json").read().strip().splitlines()[-1])
print("${w} ${name}", d["value"], d["ms_per_step"], d["stages_ms"])
PY
  done
done
