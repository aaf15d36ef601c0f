#!/bin/bash
# GPU box: A/B of the default pipeline's step time against scan workgroups per CU and contexts per GPU (dev tool).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab_stream; mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-extras --steps 400 --warmup 10"
run() { tag=$1; shift; env "$@" $B $EXTRA > $O/$tag.json 2> $O/$tag.log; python3 - <<PY
import json
try:
    j = json.load(open("$O/$tag.json")); r = j["roofline"]
    print("$tag".ljust(28), "ms/step", j["ms_per_step"], "value", j["value"], "solo", r["avg_launch_ms"], "ovl", (r.get("overlapped") or {}).get("avg_launch_ms"))
except Exception as e:
    print("$tag failed", e)
PY
}
run base X=1
run stream2 FSPANN_REFINE_STREAM=2
run stream3 FSPANN_REFINE_STREAM=3
EXTRA="--contexts 2" run ctx2 X=1
EXTRA="--contexts 4" run ctx4 X=1
EXTRA="--contexts 4" run ctx4_stream2 FSPANN_REFINE_STREAM=2
EXTRA="--contexts 2" run ctx2_stream2 FSPANN_REFINE_STREAM=2
EXTRA="--pipeline concurrent" run concurrent X=1
