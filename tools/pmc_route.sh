#!/bin/bash
# GPU box (dev tool): SQ counters of the bounded select in THROUGHPUT mode (three contexts running Route only, back to back) and
# solo (one context) — where do the wave cycles go: issuing (VALU / scalar / LDS), parked on waitcnt / barriers, or stalled at issue?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_route; rm -rf $O; mkdir -p $O
for n in 3 1; do
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  tag=n${n}_$(echo $c | cut -d' ' -f1)
  NCTXS=$n PARTS=R timeout -k 10 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$tag -- python3 $R/tools/parts_overlap.py > $O/$tag.txt 2> $O/$tag.log
done
done
python3 - <<PY
import csv, glob
O = "$O"
for f in sorted(glob.glob(f"{O}/*/*/*counter_collection.csv")):
    vals = {}
    for row in csv.DictReader(open(f)):
        if "lazy" not in row["Kernel_Name"]: continue
        vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    tag = f.split("/")[-3]
    for k, v in sorted(vals.items()):
        v = v[len(v)//4:]
        print(tag.ljust(28), k.ljust(24), len(v), round(sum(v) / len(v), 1))
PY
cat $O/*.txt
