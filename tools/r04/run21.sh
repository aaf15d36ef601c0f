#!/bin/bash
# counters of the small-tile MFMA encode at config #4's shape (1 024 / 8 192 x 1 024 x 768)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_21; rm -rf $O; mkdir -p $O
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  SHAPE=cfg4 timeout -k 10 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p$i -- python3 $R/tools/encode_bench.py > $O/p$i.txt 2> $O/p$i.log
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/p*/*/*counter_collection.csv")):
    vals=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "encode_mfma_kernel<float, 1, 1>" in r["Kernel_Name"]:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in sorted(vals.items()):
        v2=v[len(v)//4:]
        print(k.ljust(28), len(v2), round(sum(v2)/len(v2),1))
PY
