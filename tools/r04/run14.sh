#!/bin/bash
# bounded select: the crossing level's cut sized by tuples per new id — tests, phases and the step on three kinds of data
set -o pipefail
mkdir -p gpurun_out/r04_14
timeout -k 10 900 python -m pytest tests/test_gpu_route_edges.py tests/test_gpu_parity.py tests/test_gpu_tick.py tests/test_gpu_golden.py tests/test_gpu_route_fuzz.py tests/test_gpu_treeify.py tests/test_gpu_search_call.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for dk in "" siftlike:16:6 clustered; do echo "== data: $dk"; AB_LIB=tools/tmp_libs/libfspann_dbg.so DATA=$dk timeout -k 10 200 python tools/route_lazy_stamps.py 2>&1 | grep "b: cut\|total/query\|outer iters\|route kernel\|lazy" || exit 1; done
for dk in gaussian siftlike:16:6 clustered; do
  timeout -k 10 300 python bench.py --data $dk --steps 400 --warmup 20 --no-shipped --no-extras --no-cpu-baseline > gpurun_out/r04_14/headline_${dk//:/_}.json 2> gpurun_out/r04_14/headline_${dk//:/_}.err || { tail -5 gpurun_out/r04_14/headline_${dk//:/_}.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r04_14/headline_${dk//:/_}.json").read().strip().splitlines()[-1])
print("$dk", d["value"], d["ms_per_step"], d["stages_ms"], "recall@10", d.get("recall_at_10"), d["route_stage"]["kernels"][-60:])
PY
done
