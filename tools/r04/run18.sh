#!/bin/bash
# the scan kernel of the default pipeline (refine_stream_fix_kernel) alone: one context, front pipeline, kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_18; rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/front1 -- python3 $R/bench.py --no-cpu-baseline --no-extras --no-shipped --contexts 1 --steps 100 --warmup 10 > $O/front1.json 2> $O/front1.log
python3 - <<PY
import csv, glob, json
O="$O"
j=json.loads(open(f"{O}/front1.json").read().strip().splitlines()[-1])
print("front, one context: value", j["value"], "ms/step", j["ms_per_step"], "solo (events)", j["roofline"]["avg_launch_ms"])
f=glob.glob(f"{O}/front1/*/*kernel_stats.csv")[0]
for row in csv.DictReader(open(f)):
    if int(row["Calls"])>=10 and "fspann" in row["Name"] and ("refine" in row["Name"] or "front" in row["Name"] or "lazy" in row["Name"]):
        print("   ", row["Name"][:100].ljust(102), row["Calls"], round(float(row["AverageNs"])/1000,2), "us", "min", round(float(row["MinNs"])/1000,2))
PY
