#!/bin/bash
# round 4: what the driver runs at round end, rehearsed — smoke(), the driver's bench command, the same under torch.distributed.run at N = 1
set -o pipefail
mkdir -p gpurun_out/r04_final
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 1
t0=$(date +%s)
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_final/bench_driver_cmd.json 2> gpurun_out/r04_final/bench_driver_cmd.err || { tail -5 gpurun_out/r04_final/bench_driver_cmd.err; exit 1; }
echo "driver command wall: $(( $(date +%s) - t0 )) s"
timeout -k 10 900 python3 bench.py --steps 400 --warmup 20 --no-shipped > gpurun_out/r04_final/bench_400.json 2> gpurun_out/r04_final/bench_400.err || { tail -5 gpurun_out/r04_final/bench_400.err; exit 1; }
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 40 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r04_final/bench_torchrun_n1.json 2> gpurun_out/r04_final/bench_torchrun_n1.err || { tail -5 gpurun_out/r04_final/bench_torchrun_n1.err; exit 1; }
python - <<'PY'
import json
for n in ("bench_driver_cmd", "bench_400", "bench_torchrun_n1"):
    d = json.loads(open(f"gpurun_out/r04_final/{n}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(n, d["value"], d["ms_per_step"], "frac", r["frac"], "launches", r["launches"], "solo", json.dumps(r.get("solo")), "verified", d.get("all_ranks_verified"))
    if d.get("extra"):
        for k, v in d["extra"]["shipped_profiles"].items():
            print("   ", k, v.get("value"), v.get("ms_per_step"), (v.get("scan") or {}).get("frac"), v.get("route_share_of_serial_step"), v.get("error"))
    if d.get("end_to_end"):
        print("    e2e", d["end_to_end"]["value"], d["end_to_end"]["stage_ms"], json.dumps(d["end_to_end"].get("per_open"))[:400])
    if d.get("operator_surface"):
        print("    ops", d["operator_surface"]["per_query"]["route_plus_refine"], d["operator_surface"]["batched"]["queries_per_s"])
PY
