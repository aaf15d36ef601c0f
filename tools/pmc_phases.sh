#!/bin/bash
# GPU box (dev tool): instruction counts of the bounded select PHASE BY PHASE — variant builds that stop a query after phase n
# (-DFSPANN_LZ_STOP_AFTER=n, built here, four hipcc runs side by side) and the full kernel, one PMC pass each.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_phases; rm -rf $O; mkdir -p $O
FL="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -Wno-unused-function -pthread -ldl"
for n in 1 2 3 5; do (cd $R/fspann-query-system_amd && hipcc $FL -DFSPANN_LZ_STOP_AFTER=$n -o $O/lib_stop$n.so csrc/fspann_api.hip > $O/build$n.log 2>&1) & done; wait
for v in stop1 stop2 stop3 stop5 full; do
  lib=$O/lib_$v.so; [ $v = full ] && lib=
  AB_LIB=$lib NCTXS=1 PARTS=R timeout -k 10 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VMEM_RD --output-format csv -d $O/$v -- python3 $R/tools/parts_overlap.py > $O/$v.txt 2> $O/$v.log
done
python3 - <<PY
import csv, glob
O = "$O"
for v in ("stop1", "stop2", "stop3", "stop5", "full"):
    fs = glob.glob(f"{O}/{v}/*/*counter_collection.csv")
    if not fs: print(v, "no counters"); continue
    vals = {}
    for row in csv.DictReader(open(fs[0])):
        if "lazy" not in row["Kernel_Name"]: continue
        vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    w = sum(vals["SQ_WAVES"]) / len(vals["SQ_WAVES"])
    print(v.ljust(6), "per wave:", "  ".join("%s %.0f" % (k[9:], sum(x) / len(x) / w) for k, x in sorted(vals.items()) if k != "SQ_WAVES"), " (launches %d)" % len(vals["SQ_WAVES"]))
PY
