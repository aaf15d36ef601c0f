#!/bin/bash
# tiny host-pointer calls through mapped pinned memory: tests, then the per-query numbers with and without
set -o pipefail
mkdir -p gpurun_out/r04_15
timeout -k 10 900 python -m pytest tests/test_gpu_small_calls.py tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_boundary.py tests/test_gpu_cpp_host.py tests/test_gpu_refine_encode_edges.py tests/test_gpu_abi_guards.py -x -q -m gpu 2>&1 | tail -5 || exit 1
for v in "zc:" "copy:FSPANN_ZERO_COPY=0" "zc2:" "copy2:FSPANN_ZERO_COPY=0"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-shipped --no-cpu-baseline > gpurun_out/r04_15/b_${name}.json 2> gpurun_out/r04_15/b_${name}.err || { tail -5 gpurun_out/r04_15/b_${name}.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_15/b_${name}.json").read().strip().splitlines()[-1])
o=d["operator_surface"]["per_query"]
print("${name}", "enc", o["encode"], "route", o["route"], "refine", o["refine_f64_rows"], "pageable", o["refine_f64_rows_from_pageable_memory"], "r+r", o["route_plus_refine"], "batched", d["operator_surface"]["batched"]["queries_per_s"])
PY
done
