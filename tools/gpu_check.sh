#!/bin/bash
# One gpurun call: GPU test-suite, then the bench lines used while developing (outputs under gpurun_out/).
# usage: bash tools/gpu_check.sh [pytest-args...]   (default: the whole -m gpu suite)
set -o pipefail
mkdir -p gpurun_out
if [ "$#" -gt 0 ]; then sel=("$@"); else sel=(tests); fi
timeout -k 10 1000 python -m pytest "${sel[@]}" -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -n 15 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
for cfg in "" "--tasks 20 --steps 10"; do
  tag=$(echo "$cfg" | tr -d ' -')
  timeout -k 10 300 python bench.py --no-cpu $cfg > gpurun_out/bench_$tag.log 2>&1 || { tail -n 5 gpurun_out/bench_$tag.log; exit 1; }
  python - "gpurun_out/bench_$tag.log" "$cfg" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(f"[{sys.argv[2]}]", d["ms_per_step"], d["kernels_ms"], "path", d["path_roofline_frac"], "floor", d["two_pass_floor_frac"], d["roofline"]["peak_measured"])
PY
done
