#!/bin/bash
# dev A/B: the end-to-end host pipeline (decrypt stage) of bench.py, base library vs current, same box; plus the open micro-benchmark
set -o pipefail
mkdir -p gpurun_out/r04_e2e
g++ -O2 -std=c++17 -pthread tools/micro/open_bench.cpp -ldl -o /tmp/open_bench && /tmp/open_bench 262144 16 | tee gpurun_out/r04_e2e/open_bench.txt
for v in base=tools/tmp_libs/lib_base.so cur=fspann-query-system_amd/libfspann_hip.so; do
  name=${v%%=*}; lib=${v#*=}
  AB_LIB=$lib timeout -k 10 400 python tools/ab_bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-shipped > gpurun_out/r04_e2e/$name.json 2> gpurun_out/r04_e2e/$name.err || { tail -5 gpurun_out/r04_e2e/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_e2e/$name.json").read().strip().splitlines()[-1])
print("$name", json.dumps(d["end_to_end"]["stage_ms"]), d["end_to_end"]["value"], d["end_to_end"]["matches_kernel_path"])
PY
done
timeout -k 10 600 python -m pytest tests/test_gpu_hostpipe.py tests/test_gpu_rotate_migrate.py -x -q -m gpu 2>&1 | tail -3
