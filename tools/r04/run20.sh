#!/bin/bash
# kernel-level split of the MFMA encode path at the shapes of BASELINE configs #3 / #4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_20; rm -rf $O; mkdir -p $O
for s in cfg3 cfg4; do
  SHAPE=$s timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$s -- python3 $R/tools/encode_bench.py > $O/$s.txt 2> $O/$s.log
  python3 - <<PY
import csv, glob
f=glob.glob("$O/$s/*/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "encode" in r["Kernel_Name"] or "fill" in r["Kernel_Name"].lower()]
import collections
agg=collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"][:60], r["Grid_Size"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000)
for k,v in sorted(agg.items()):
    v2=v[len(v)//4:]
    print("$s", k[0].ljust(62), "grid", k[1].rjust(9), "n", len(v), "avg us", round(sum(v2)/len(v2),1))
PY
done
