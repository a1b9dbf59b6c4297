#!/bin/bash
# Pass times against the rows per work unit (power-of-two against odd multiples of the 256-row block), six candidate
# output allocations each: are the slow placement levels a matter of every wave's streams being 32 KiB-congruent?
set -o pipefail
mkdir -p gpurun_out
for ur in ${URS:-0 7936 8448 6912 5888 9984}; do
  timeout -k 10 300 python bench.py --no-cpu --placement-candidates 6 --unit-rows $ur $EXTRA > gpurun_out/abu_$ur.log 2>&1 || { tail -n 5 gpurun_out/abu_$ur.log; exit 1; }
  python - "gpurun_out/abu_$ur.log" "$ur" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(f"[unit_rows {sys.argv[2]}]", d["ms_per_step"], d["kernels_ms"]["k_gram"], d["kernels_ms"]["k_basis_project"], d["config"]["output_placement"][-58:])
PY
done
