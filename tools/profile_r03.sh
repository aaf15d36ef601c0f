#!/bin/bash
# GPU box: the rocprofv3 summaries committed under profiles/ (kernel trace + stats; PMC traffic in separate passes).
# usage (through gpurun): bash tools/profile_r03.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03_v1}; O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
B="$R/bench.py --no-cpu-baseline --no-extras"
# 1) the default command (concurrent contexts): what the driver's line is produced by
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/default -- python3 $B --steps 100 --warmup 10 > $O/default.json 2> $O/default.log
# 2) one context, one stream: every kernel's duration is a solo duration
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 $B --pipeline serial --steps 50 --warmup 5 > $O/serial.json 2> $O/serial.log
# 3) PMC, separate passes, serial pipeline (counters per dispatch)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $B --pipeline serial --steps 10 --warmup 2 > $O/pmc_$c.json 2> $O/pmc_$c.log
done
python3 - <<PY
import csv, glob, json, os
O = "$O"
for name in ("default", "serial"):
    try:
        j = json.loads(open(f"{O}/{name}.json").read())
        r = j["roofline"]
        print(name, "value", j["value"], "ms/step", j["ms_per_step"], "solo", r["avg_launch_ms"], "frac", r["frac"], "overlapped", (r.get("overlapped") or {}).get("avg_launch_ms"))
        f = glob.glob(f"{O}/{name}/*/*kernel_stats.csv")[0]
        for row in csv.DictReader(open(f)):
            if int(row["Calls"]) >= 20:
                print("   ", row["Name"][:80].ljust(82), row["Calls"], round(float(row["AverageNs"]) / 1000, 2), "us")
    except Exception as e:
        print(name, "failed:", e)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    try:
        f = glob.glob(f"{O}/pmc_{c}/*/*counter_collection.csv")[0]
        vals = {}
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == c:
                vals.setdefault(row["Kernel_Name"][:60], []).append(float(row["Counter_Value"]))
        for k, v in vals.items():
            if len(v) >= 5:
                print(c, k.ljust(62), len(v), round(sum(v) / len(v), 1))
    except Exception as e:
        print(c, "failed:", e)
PY
# ---- round 3 additions ------------------------------------------------------------------------------------------------------
# 4) the reference's shipped profiles (config_sift1m.json), one context, one stream: solo durations of the full select, the chunked
#    scan and the merge
for w in sift1m_P4_FAST sift1m_P10_HIGH; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -- python3 $B --workload $w --data clustered --pipeline serial --steps 20 --warmup 3 > $O/$w.json 2> $O/$w.log
done
# 5) MFMA counters of the coding kernel (Setup runs inside every bench process): a separate PMC pass, kernel trace only beside it
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 $B --pipeline serial --steps 4 --warmup 1 > $O/pmc_mfma.json 2> $O/pmc_mfma.log
python3 - <<PY
import csv, glob, json
O = "$O"
for name in ("sift1m_P4_FAST", "sift1m_P10_HIGH"):
    try:
        j = json.loads(open(f"{O}/{name}.json").read())
        print(name, "value", j["value"], "ms/step", j["ms_per_step"], "scan solo", j["roofline"]["avg_launch_ms"], "frac", j["roofline"]["frac"])
        for row in csv.DictReader(open(glob.glob(f"{O}/{name}/*/*kernel_stats.csv")[0])):
            if int(row["Calls"]) >= 10 and "fspann" in row["Name"]:
                print("   ", row["Name"][:80].ljust(82), row["Calls"], round(float(row["AverageNs"]) / 1000, 2), "us")
    except Exception as e:
        print(name, "failed:", e)
try:
    vals = {}
    for row in csv.DictReader(open(glob.glob(f"{O}/pmc_mfma/*/*counter_collection.csv")[0])):
        if "encode_mfma" in row["Kernel_Name"]:
            vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for k, v in sorted(vals.items()):
        print("encode_mfma_kernel", k.ljust(28), len(v), round(sum(v) / len(v), 1))
except Exception as e:
    print("pmc_mfma failed:", e)
PY
