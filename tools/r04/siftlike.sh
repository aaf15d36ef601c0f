#!/bin/bash
# recall of the shipped profiles on SIFT-like synthetic data (intrinsic dimension r + noise)
set -o pipefail
mkdir -p gpurun_out/r04_siftlike
for w in sift1m_P4_FAST sift1m_P10_HIGH; do
  for dk in "siftlike:16:6" "siftlike:12:6" "siftlike:24:6" "siftlike:16:16"; do
    name=${dk//:/_}
    timeout -k 10 300 python bench.py --workload $w --k 100 --data $dk --steps 20 --warmup 3 --prewarm 5 --no-extras --no-cpu-baseline --no-shipped --solo-tail 0 > gpurun_out/r04_siftlike/${w}_${name}.json 2> gpurun_out/r04_siftlike/${w}_${name}.err || { tail -5 gpurun_out/r04_siftlike/${w}_${name}.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/r04_siftlike/${w}_${name}.json").read().strip().splitlines()[-1])
print("${w} ${dk}", d["value"], d["ms_per_step"], "recall@10", d.get("recall_at_10"), "recall@k", d.get("recall_at_k"), "ratio", d.get("distance_ratio_at_10"), d["treeified"])
PY
  done
done
