#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu "$@" > gpurun_out/r03_bench_$tag.log 2>&1 || { tail -n 8 gpurun_out/r03_bench_$tag.log; return 1; }
  python - gpurun_out/r03_bench_$tag.log "$tag" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(f"[{sys.argv[2]}]", d["ms_per_step"], d.get("untuned",{}).get("ms_per_step"), d["kernels_ms"], "path", d.get("path_roofline_frac"), [r["kernels_ms"] for r in d.get("per_rank",{}).get("ranks",[])])
PY
}
run u4096 && run u2048 --unit-rows 2048 && run n20 --tasks 20 --steps 10 && run n20_u2048 --tasks 20 --steps 10 --unit-rows 2048 && run b32_u2048 --model ViT-B-32 --unit-rows 2048 || exit 1
SVDQ_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --no-cpu --steps 5 --no-weak > gpurun_out/r03_bench_gloo2.log 2>&1; tail -n 2 gpurun_out/r03_bench_gloo2.log | cut -c1-300; grep -o '"per_rank".*' gpurun_out/r03_bench_gloo2.log | cut -c1-500
timeout -k 10 200 python tools/host_overhead.py > gpurun_out/r03_host_overhead.log 2>&1; head -n 6 gpurun_out/r03_host_overhead.log
