#!/bin/bash
# The other bench configurations in one call (each its own process): masked, from-base, ViT-B-32, N = 2/4/12/16
set -o pipefail
mkdir -p gpurun_out
while read -r cfg; do
  tag=$(echo "$cfg" | tr -d ' -')
  timeout -k 10 300 python bench.py --no-cpu $cfg > gpurun_out/bm_$tag.log 2>&1 || { tail -n 5 gpurun_out/bm_$tag.log; exit 1; }
  python - "gpurun_out/bm_$tag.log" "$cfg" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(f"[{sys.argv[2]}]", d["ms_per_step"], round(d["value"]/1e3), "k", d["kernels_ms"], "path", d["path_roofline_frac"], "untuned", d.get("untuned", {}).get("ms_per_step"))
PY
done <<'CFGS'
--model ViT-B-32
--model ViT-B-16 --masks union --stages 4
--model ViT-L-14 --masks union
--from-base fused
--tasks 2
--tasks 4
--tasks 12
--tasks 16
CFGS
