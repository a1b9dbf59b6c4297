#!/bin/bash
set -o pipefail
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu "$@" > gpurun_out/r03_bench_$tag.log 2>&1 || { tail -n 8 gpurun_out/r03_bench_$tag.log; return 1; }
  python - gpurun_out/r03_bench_$tag.log "$tag" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(f"[{sys.argv[2]}]", d["ms_per_step"], d.get("untuned",{}).get("ms_per_step"), d["kernels_ms"], "path", d.get("path_roofline_frac"))
PY
}
mkdir -p gpurun_out
run n20 --tasks 20 --steps 10 && run n20_u8192 --tasks 20 --steps 10 --unit-rows 8192 && run n20_u16384 --tasks 20 --steps 10 --unit-rows 16384 || exit 1
bash tools/r03_pmc_traffic.sh n8 && bash tools/r03_pmc_traffic.sh b16_walk --model ViT-B-16 --masks union --stages 4 && bash tools/r03_pmc_traffic.sh n20 --tasks 20
