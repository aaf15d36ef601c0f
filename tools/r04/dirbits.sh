#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04_dirbits
for b in 6 4 2 0; do
  echo "== FSPANN_ROUTE_DIR_EXTRA_BITS=$b"
  FSPANN_ROUTE_DIR_EXTRA_BITS=$b AB_LIB=tools/tmp_libs/libfspann_dbg.so timeout -k 10 300 python tools/route_probe_stamps.py 2>&1 | grep "probes with"
  for pipe in front serial; do
    FSPANN_ROUTE_DIR_EXTRA_BITS=$b timeout -k 10 300 python bench.py --steps 400 --warmup 20 --no-extras --no-cpu-baseline --no-shipped --pipeline $pipe > gpurun_out/r04_dirbits/bench_${b}_${pipe}.json 2> gpurun_out/r04_dirbits/bench_${b}_${pipe}.err || { tail -5 gpurun_out/r04_dirbits/bench_${b}_${pipe}.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/r04_dirbits/bench_${b}_${pipe}.json").read().strip().splitlines()[-1])
print("bits+${b} ${pipe}", d["value"], d["ms_per_step"], d["stages_ms"]["route_select"])
PY
  done
done
