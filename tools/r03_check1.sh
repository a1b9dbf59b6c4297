#!/bin/bash
# round 3, first GPU call: the new walk-mode tests first (fast failure), the whole GPU suite, then bench lines and probes
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "walk or combine_starts or gather_mode" > gpurun_out/r03_walk_tests.log 2>&1
rc=$?; tail -n 25 gpurun_out/r03_walk_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -n 15 gpurun_out/pytest_gpu.log; [ $rc -ne 0 ] && exit $rc
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-cpu "$@" > gpurun_out/r03_bench_$tag.log 2>&1 || { tail -n 8 gpurun_out/r03_bench_$tag.log; return 1; }
  python - gpurun_out/r03_bench_$tag.log "$tag" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(f"[{sys.argv[2]}]", d["ms_per_step"], d.get("untuned",{}).get("ms_per_step"), d["kernels_ms"], "path", d["path_roofline_frac"], "dens", d["config"].get("mask_density"))
PY
}
run default && run b16 --model ViT-B-16 --stages 4 && run b16_walk --model ViT-B-16 --masks union --stages 4 && \
run b16_index --model ViT-B-16 --masks union --stages 4 --masks-index && run b16_walk_packed --model ViT-B-16 --masks union --stages 4 --masks-packed && \
run b32 --model ViT-B-32 || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -Wno-unused-value --offload-arch=gfx950 -I svd-quantization-task-merging_amd/csrc tools/probe/eig_time.hip -o /tmp/eig_time && \
  { /tmp/eig_time 8 296; /tmp/eig_time 8 13; /tmp/eig_time 20 296; } > gpurun_out/r03_eig_time.log 2>&1; cat gpurun_out/r03_eig_time.log
timeout -k 10 200 python tools/shard_one.py > gpurun_out/r03_shard_one.log 2>&1; tail -n 6 gpurun_out/r03_shard_one.log
