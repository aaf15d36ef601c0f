#!/bin/bash
# After `tools/gr.sh -- 'bash tools/profile_r02.sh <tag>'`: copy the summaries of gpurun_out/prof_<tag>/ into profiles/ (tracked).
# usage: bash tools/profile_collect.sh <tag>
set -e
cd "$(dirname "$0")/.."
TAG=$1; O=gpurun_out/prof_$TAG
for name in default serial; do
  cp $(ls $O/$name/*/*kernel_stats.csv | head -1) profiles/${TAG}_${name}_kernel_stats.csv
  cp $O/$name.json profiles/${TAG}_${name}_bench_under_rocprof.json
done
for c in FETCH_SIZE WRITE_SIZE; do
  python3 - "$O" "$c" "$TAG" <<'PY'
import csv, glob, sys
O, c, tag = sys.argv[1:]
f = glob.glob(f"{O}/pmc_{c}/*/*counter_collection.csv")[0]
rows = list(csv.DictReader(open(f)))
keep = [r for r in rows if "fspann::" in r["Kernel_Name"]]
cols = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Counter_Name", "Counter_Value"]
w = csv.DictWriter(open(f"profiles/{tag}_pmc_{c}.csv", "w", newline=""), fieldnames=cols, extrasaction="ignore")
w.writeheader()
w.writerows(keep)
print(c, len(keep), "rows")
PY
done
python3 tools/pmc_traffic.py $TAG
