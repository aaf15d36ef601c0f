#!/bin/bash
# round 4, GPU call 2: the driver's command with the new extras (time it), output kept
set -o pipefail
mkdir -p gpurun_out/r04_2
t0=$(date +%s)
timeout -k 10 1000 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_2/bench.json 2> gpurun_out/r04_2/bench.err
rc=$?
echo "rc=$rc wall=$(( $(date +%s) - t0 ))s"
tail -5 gpurun_out/r04_2/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_2/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], json.dumps(d["roofline"]["solo"]), d["roofline"]["frac"], d["roofline"]["launches"])
print(json.dumps(d["operator_surface"], indent=1))
print(json.dumps(d["encode_stage"]["query_side_mfma_vs_exact"], indent=1))
print(json.dumps(d["roofline"]["f64_rows"]))
print(json.dumps(d["extra"], indent=1)[:6000])
print(json.dumps(d["cpu_baseline"]))
PY
