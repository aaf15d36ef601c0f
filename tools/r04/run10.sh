#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r04_10
timeout -k 10 900 python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_route_edges.py tests/test_gpu_refine_encode_edges.py tests/test_gpu_treeify.py tests/test_gpu_abi_guards.py tests/test_gpu_cpp_host.py -x -q -m gpu 2>&1 | tail -4 || exit 1
timeout -k 10 600 python bench.py --steps 100 --warmup 5 --no-shipped --no-cpu-baseline > gpurun_out/r04_10/bench.json 2> gpurun_out/r04_10/bench.err || { tail -5 gpurun_out/r04_10/bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_10/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
print(json.dumps(d["operator_surface"], indent=1))
PY
