#!/bin/bash
# dev A/B on one box: bench.py at a workload against variant libraries.  usage: tools/r04/ab.sh <outdir> <workload args...> -- name=lib[:ENV=..] ...
set -o pipefail
out=gpurun_out/$1; shift; mkdir -p $out
args=()
while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
for v in "$@"; do
  name=${v%%=*}; rest=${v#*=}; lib=${rest%%:*}; envs=""; [ "$rest" != "$lib" ] && envs=${rest#*:}
  env AB_LIB=$lib $envs timeout -k 10 400 python tools/ab_bench.py "${args[@]}" > $out/$name.json 2> $out/$name.err || { tail -5 $out/$name.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$out/$name.json").read().strip().splitlines()[-1])
print("$name", d["value"], d["ms_per_step"], d["stages_ms"], (d.get("roofline") or {}).get("frac"))
PY
done
