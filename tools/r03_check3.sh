#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -n 6 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu "$@" > gpurun_out/r03_bench_$tag.log 2>&1 || { tail -n 8 gpurun_out/r03_bench_$tag.log; return 1; }
  python - gpurun_out/r03_bench_$tag.log "$tag" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(f"[{sys.argv[2]}]", d["ms_per_step"], d.get("untuned",{}).get("ms_per_step"), d["kernels_ms"], "path", d.get("path_roofline_frac"))
PY
}
run default && run n20 --tasks 20 --steps 10 && run n12 --tasks 12 --steps 10 && run n16 --tasks 16 --steps 10 && run n24 --tasks 24 --steps 8 && run n4 --tasks 4 && run b32 --model ViT-B-32 && run b16_walk --model ViT-B-16 --masks union --stages 4
