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
# round 3: the shipped profiles' traces and the MFMA counters of the coding kernel
for w in sift1m_P4_FAST sift1m_P10_HIGH; do
  if ls $O/$w/*/*kernel_stats.csv > /dev/null 2>&1; then
    cp $(ls $O/$w/*/*kernel_stats.csv | head -1) profiles/${TAG}_${w}_serial_kernel_stats.csv
    cp $O/$w.json profiles/${TAG}_${w}_serial_bench_under_rocprof.json
  fi
done
if ls $O/pmc_mfma/*/*counter_collection.csv > /dev/null 2>&1; then
  python3 - "$O" "$TAG" <<'PY'
import csv, glob, sys
O, tag = sys.argv[1:]
rows = [r for r in csv.DictReader(open(glob.glob(f"{O}/pmc_mfma/*/*counter_collection.csv")[0])) if "encode_" in r["Kernel_Name"]]
cols = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Counter_Name", "Counter_Value"]
w = csv.DictWriter(open(f"profiles/{tag}_pmc_mfma_encode.csv", "w", newline=""), fieldnames=cols, extrasaction="ignore")
w.writeheader(); w.writerows(rows)
print("pmc_mfma", len(rows), "rows")
PY
fi
