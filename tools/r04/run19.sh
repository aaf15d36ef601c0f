#!/bin/bash
# one GPU's shard of BASELINE configs #3 (GIST-shaped) and #4 (10 M x 768): bench lines + kernel traces
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_19; rm -rf $O; mkdir -p $O
for w in gist1m_T16_b32_B512_Q512 synth10m_T32_b64_B1024_Q1024; do
  echo "== $w"; date +%T
  timeout -k 10 900 python3 $R/bench.py --workload $w --steps 60 --warmup 5 --no-extras --no-shipped --no-cpu-baseline > $O/$w.json 2> $O/$w.err || { tail -8 $O/$w.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("$O/$w.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$w", d["value"], d["ms_per_step"], d["stages_ms"], "frac", r["frac"], r.get("kernel"), "solo ms", r["avg_launch_ms"], "recall", d.get("recall_at_10"), d["config"]["pipeline"][:60], d["route_stage"]["kernels"][:80])
PY
  date +%T
done
