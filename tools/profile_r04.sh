#!/bin/bash
# GPU box: the rocprofv3 summaries committed under profiles/ (kernel trace + stats; PMC in separate passes).
# usage (through gpurun): bash tools/profile_r04.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r04_v1}; O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
B="$R/bench.py --no-cpu-baseline --no-extras --no-shipped"
# 1) the default command's pipeline (three contexts, front launch + scan): what the driver's line is produced by
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/default -- python3 $B --steps 100 --warmup 10 > $O/default.json 2> $O/default.log
echo "default done"
# 2) one context, one stream: every kernel's duration is a solo duration
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 $B --pipeline serial --steps 50 --warmup 5 > $O/serial.json 2> $O/serial.log
echo "serial done"
# 3) PMC, separate passes, serial pipeline (counters per dispatch)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $B --pipeline serial --steps 10 --warmup 2 > $O/pmc_$c.json 2> $O/pmc_$c.log
  echo "pmc $c done"
done
# 4) the reference's shipped profiles, one context, one stream: solo durations of the full select and the run-of-chunks scan
for w in sift1m_P4_FAST sift1m_P10_HIGH; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -- python3 $B --workload $w --k 100 --data ${SHIPPED_DATA:-siftlike:16:6} --pipeline serial --steps 20 --warmup 3 > $O/$w.json 2> $O/$w.log
  echo "$w done"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${w}_pmc_$c -- python3 $B --workload $w --k 100 --data ${SHIPPED_DATA:-siftlike:16:6} --pipeline serial --steps 6 --warmup 2 > $O/${w}_pmc_$c.json 2> $O/${w}_pmc_$c.log
  done
  echo "$w pmc done"
done
# 5) query-side encode, exact vs MFMA pre-filter, with the MFMA busy counters (a separate PMC pass)
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_encode_q -- python3 $R/tools/encode_q_pmc.py > $O/pmc_encode_q.txt 2> $O/pmc_encode_q.log
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/encode_q -- python3 $R/tools/encode_q_pmc.py > $O/encode_q.txt 2> $O/encode_q.log
echo "encode done"
# 6) bounded select: SQ counters (solo, one context running Route only) — LDS bank conflicts, wait share, instruction mix
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
         "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  tag=route_$(echo $c | cut -d' ' -f1)
  NCTXS=1 PARTS=R timeout -k 10 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$tag -- python3 $R/tools/parts_overlap.py > $O/$tag.txt 2> $O/$tag.log
done
echo "route pmc done"
python3 - <<PY
import csv, glob, json, os
O = "$O"
def stats(name, min_calls=10):
    try:
        j = json.loads(open(f"{O}/{name}.json").read().strip().splitlines()[-1])
        r = j["roofline"]
        print(name, "value", j["value"], "ms/step", j["ms_per_step"], "solo", r["avg_launch_ms"], "frac", r["frac"], "launches", r["launches"])
        f = glob.glob(f"{O}/{name}/*/*kernel_stats.csv")[0]
        for row in csv.DictReader(open(f)):
            if int(row["Calls"]) >= min_calls and "fspann" in row["Name"]:
                print("   ", row["Name"][:90].ljust(92), row["Calls"], round(float(row["AverageNs"]) / 1000, 2), "us")
    except Exception as e:
        print(name, "failed:", e)
for nm in ("default", "serial", "sift1m_P4_FAST", "sift1m_P10_HIGH"):
    stats(nm)
for f in sorted(glob.glob(f"{O}/*pmc_*/*/*counter_collection.csv")) + sorted(glob.glob(f"{O}/route_*/*/*counter_collection.csv")):
    vals = {}
    for row in csv.DictReader(open(f)):
        vals.setdefault((row["Kernel_Name"][:50], row["Counter_Name"]), []).append(float(row["Counter_Value"]))
    tag = f.split("/")[-3]
    for (k, cn), v in sorted(vals.items()):
        if len(v) >= 4 and ("refine_" in k or "lazy" in k or "route_select" in k or "encode_" in k):
            v2 = v[len(v) // 4:]
            print(tag.ljust(30), k.ljust(52), cn.ljust(26), len(v2), round(sum(v2) / len(v2), 1))
try:
    f = glob.glob(f"{O}/encode_q/*/*kernel_stats.csv")[0]
    for row in csv.DictReader(open(f)):
        if "encode" in row["Name"]:
            print("encode_q", row["Name"][:80].ljust(82), row["Calls"], round(float(row["AverageNs"]) / 1000, 2), "us")
except Exception as e:
    print("encode_q failed:", e)
PY
