cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_sq; rm -rf $O; mkdir -p $O
for c in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$tag -- python3 $R/bench.py --no-cpu-baseline --no-extras --pipeline serial --steps 10 --warmup 2 > $O/$tag.json 2> $O/$tag.log
done
python3 - <<PY
import csv, glob
O = "$O"
for f in sorted(glob.glob(f"{O}/*/*/*counter_collection.csv")):
    vals = {}
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"][:48], row["Counter_Name"])
        vals.setdefault(k, []).append(float(row["Counter_Value"]))
    for k, v in sorted(vals.items()):
        if len(v) >= 5 and "fspann" in k[0] and ("lazy" in k[0] or "refine_stream" in k[0] or "encode_exact" in k[0]):
            print(k[1].ljust(22), k[0].ljust(50), len(v), round(sum(v) / len(v), 1))
PY
