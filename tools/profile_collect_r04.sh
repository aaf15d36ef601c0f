#!/bin/bash
# After `gpurun -- 'bash tools/profile_r04.sh <tag>'`: copy the summaries of gpurun_out/prof_<tag>/ into profiles/ (tracked).
# usage: bash tools/profile_collect_r04.sh <tag>
set -e
cd "$(dirname "$0")/.."
TAG=$1; O=gpurun_out/prof_$TAG
for name in default serial; do
  cp $(ls $O/$name/*/*kernel_stats.csv | head -1) profiles/${TAG}_${name}_kernel_stats.csv
  cp $O/$name.json profiles/${TAG}_${name}_bench_under_rocprof.json
done
for w in sift1m_P4_FAST sift1m_P10_HIGH; do
  cp $(ls $O/$w/*/*kernel_stats.csv | head -1) profiles/${TAG}_${w}_serial_kernel_stats.csv
  cp $O/$w.json profiles/${TAG}_${w}_serial_bench_under_rocprof.json
done
cp $(ls $O/encode_q/*/*kernel_stats.csv | head -1) profiles/${TAG}_encode_query_side_kernel_stats.csv
python3 - "$O" "$TAG" <<'PY'
import csv, glob, sys
O, tag = sys.argv[1:]
cols = ["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Counter_Name", "Counter_Value"]
def keep(src_glob, dst, pred):
    rows = []
    for f in sorted(glob.glob(src_glob)):
        rows += [r for r in csv.DictReader(open(f)) if pred(r["Kernel_Name"])]
    w = csv.DictWriter(open(f"profiles/{tag}_{dst}.csv", "w", newline=""), fieldnames=cols, extrasaction="ignore")
    w.writeheader(); w.writerows(rows)
    print(dst, len(rows), "rows")
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    keep(f"{O}/pmc_{c}/*/*counter_collection.csv", f"pmc_{c}", lambda k: "fspann::" in k and ("refine_" in k or "route_" in k or "encode_exact_kernel<float, 4>" in k))
    for w in ("sift1m_P4_FAST", "sift1m_P10_HIGH"):
        keep(f"{O}/{w}_pmc_{c}/*/*counter_collection.csv", f"{w}_pmc_{c}", lambda k: "refine_" in k or "route_" in k)
keep(f"{O}/pmc_encode_q/*/*counter_collection.csv", "pmc_mfma_encode_query_side", lambda k: "encode_" in k)
keep(f"{O}/route_*/*/*counter_collection.csv", "pmc_route_sq", lambda k: "route_select_lazy" in k)
PY
python3 tools/pmc_traffic.py $TAG
cp gpurun_out/profile_${TAG}.txt profiles/${TAG}_summary.txt 2>/dev/null || true
ls -la profiles | grep $TAG
