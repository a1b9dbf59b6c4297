#!/bin/bash
# tune_placement (candidates walked through the memory regions, basis and mean chosen separately) against the first allocation
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_parity.py -q -x -m gpu -k tune_placement > gpurun_out/t_place.log 2>&1 || { tail -n 20 gpurun_out/t_place.log; exit 1; }
for cfg in "--placement-candidates 1" "--placement-candidates 4" "--placement-candidates 6" "--tasks 20 --steps 8 --placement-candidates 1" "--tasks 20 --steps 8 --placement-candidates 4"; do
  tag=$(echo "$cfg" | tr -d ' -')
  timeout -k 10 300 python bench.py --no-cpu $cfg > gpurun_out/abp_$tag.log 2>&1 || { tail -n 5 gpurun_out/abp_$tag.log; exit 1; }
  python - "gpurun_out/abp_$tag.log" "$cfg" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(f"[{sys.argv[2]}]", d["ms_per_step"], d["kernels_ms"]["k_gram"], d["kernels_ms"]["k_basis_project"], d["config"]["output_placement"][-90:])
PY
done
