#!/bin/bash
# full select on data whose candidate lists overlap little: LDS-staged repeats + groups sized by the occupied levels
set -o pipefail
mkdir -p gpurun_out/r04_11
timeout -k 10 900 python -m pytest tests/test_gpu_shipped_profiles.py tests/test_gpu_treeify.py tests/test_gpu_route_edges.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_route_fuzz.py tests/test_gpu_search_call.py -x -q -m gpu > gpurun_out/r04_11/tests.log 2>&1
rc=$?
tail -5 gpurun_out/r04_11/tests.log
[ $rc -ne 0 ] && exit $rc
export AB_LIB=tools/tmp_libs/libfspann_dbg.so
for p in P4 P10; do
  echo "== $p clustered"; timeout -k 10 200 python tools/route_full_stamps.py $p || exit 1
  echo "== $p siftlike"; DATA=siftlike:16:6 timeout -k 10 200 python tools/route_full_stamps.py $p || exit 1
done > gpurun_out/r04_11/stamps.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04_11/stamps.txt
