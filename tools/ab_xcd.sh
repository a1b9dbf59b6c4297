#!/bin/bash
# A/B of the unit order (interleaved over the XCDs vs XCD-chunked), six candidate output allocations each
set -o pipefail
mkdir -p gpurun_out
for cfg in "" "--xcd" "--tasks 20 --steps 8" "--tasks 20 --steps 8 --xcd" "--model ViT-B-32" "--model ViT-B-32 --xcd"; do
  tag=$(echo "$cfg" | tr -d ' -')
  timeout -k 10 300 python bench.py --no-cpu --placement-candidates 6 $cfg > gpurun_out/abx_$tag.log 2>&1 || { tail -n 5 gpurun_out/abx_$tag.log; exit 1; }
  python - "gpurun_out/abx_$tag.log" "$cfg" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(f"[{sys.argv[2]}]", d["ms_per_step"], d["kernels_ms"], d["config"]["output_placement"][-60:])
PY
done
