#!/bin/bash
# Dev tool (GPU box): per-kernel average durations of one bench run under rocprofv3.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kstats; rm -rf $O; mkdir -p $O
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --no-cpu-baseline --no-pipelined --steps 30 --warmup 5 "$@" > $O/bench.json 2> $O/log.txt
python3 - <<PY
import csv, glob, json
j = json.loads(open("$O/bench.json").read())
print("value", j["value"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"], "bracket", j["roofline"]["avg_launch_ms"])
for r in csv.DictReader(open(glob.glob("$O/*/*kernel_stats.csv")[0])):
    if int(r["Calls"]) >= 30: print(r["Name"][:70].ljust(72), r["Calls"], round(float(r["AverageNs"]) / 1000, 2), "us")
PY
