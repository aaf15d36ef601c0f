"""profiles/refine_traffic.json from the two PMC passes of tools/profile_r02.sh (FETCH_SIZE, WRITE_SIZE; separate rocprofv3 runs).

usage: python tools/pmc_traffic.py <tag>      (reads profiles/<tag>_pmc_FETCH_SIZE.csv and profiles/<tag>_pmc_WRITE_SIZE.csv)

Units and correction as /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: both counters are in KB
(1024 B); on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B / lane) coalesced read -> x2; WRITE_SIZE exact."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
KERNEL = "refine_stream_kernel<float, float, 32, false"      # (round 4: ..., false, false> — the template gained the runs-of-chunks switch)


def avg(counter):
    vals = []
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_{counter}.csv")) as f:
        for row in csv.DictReader(f):
            if KERNEL in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    if len(vals) < 5:
        raise SystemExit(f"{counter}: only {len(vals)} dispatches of {KERNEL}")
    vals = vals[2:]          # the warm-up steps
    return sum(vals) / len(vals), len(vals)


fetch_kb, n = avg("FETCH_SIZE")
write_kb, _ = avg("WRITE_SIZE")
Q, B, d, k = 1024, 256, 128, 10
alg = Q * (B * d * 4 + d * 4 + k * 8)
path = os.path.join(ROOT, "profiles", "refine_traffic.json")
j = json.load(open(path)) if os.path.exists(path) else {}
j["dense"] = {
    "workload": "sift1m_T16_b32_B256_Q1024", "Q": Q, "kernel": "refine_stream_kernel<float,float,32,false,false>",
    "hbm_bytes_per_launch": int(round((2 * fetch_kb + write_kb) * 1024)),
    "FETCH_SIZE_KB_avg": round(fetch_kb, 1), "WRITE_SIZE_KB_avg": round(write_kb, 1), "launches": n,
    "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of a wide coalesced (16 B/lane) streaming read -> x2 "
                  "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; units KB = 1024 B",
    "source": f"profiles/{tag}_pmc_FETCH_SIZE.csv, profiles/{tag}_pmc_WRITE_SIZE.csv (separate rocprofv3 --kernel-trace --pmc passes of "
              "`python3 bench.py --no-cpu-baseline --no-extras --no-shipped --pipeline serial --steps 10 --warmup 2`; 32 dense blocks = 4.3 GB cycled, "
              "so the rows come from HBM)",
    "algorithmic_bytes_per_launch": alg,
}
json.dump(j, open(path, "w"), indent=1)
print("dense:", j["dense"]["hbm_bytes_per_launch"], "B per launch =", round(j["dense"]["hbm_bytes_per_launch"] / alg, 3), "x algorithmic")
