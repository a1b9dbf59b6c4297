#!/bin/bash
# Round-3 profile collection on the GPU box (run from the repo root): everything profiles/r03_* is made from.  Every
# rocprofv3 invocation runs once; counter passes use --pmc with --kernel-trace only.  Outputs land in gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out
# 1. the default bench line (with the CPU baseline), as the driver runs it
timeout -k 10 400 python bench.py > $O/r03_bench_default.log 2>&1 || { tail -n 5 $O/r03_bench_default.log; exit 1; }
grep '^{' $O/r03_bench_default.log > $O/r03_bench_line.json; cut -c1-260 $O/r03_bench_line.json
# 2. kernel statistics + timed-region averages: headline, masked (walk and index lists), N = 20, merge
bash tools/r03_prof1.sh bench || exit 1
bash tools/r03_prof1.sh masked_vitb16 --model ViT-B-16 --masks union --stages 4 || exit 1
bash tools/r03_prof1.sh masked_vitb16_index --model ViT-B-16 --masks union --stages 4 --masks-index || exit 1
bash tools/r03_prof1.sh n20 --tasks 20 || exit 1
bash tools/r03_prof1.sh merge --merge || exit 1
bash tools/r03_prof1.sh merge_n20c2 --merge --tasks 20 --clusters 2 || exit 1
# 3. HBM traffic (FETCH_SIZE, WRITE_SIZE), one pass each
bash tools/r03_pmc_traffic.sh n8 && bash tools/r03_pmc_traffic.sh masked_vitb16 --model ViT-B-16 --masks union --stages 4 && \
  bash tools/r03_pmc_traffic.sh n20 --tasks 20 || exit 1
# 4. SQ counters, two passes each
bash tools/r03_pmc_sq.sh n8 > /dev/null && bash tools/r03_pmc_sq.sh n20 --tasks 20 > /dev/null && \
  bash tools/r03_pmc_sq.sh masked_vitb16 --model ViT-B-16 --masks union --stages 4 > /dev/null || exit 1
# 5. other bench lines and probes
for cfg in "b32 --model ViT-B-32" "b16 --model ViT-B-16 --stages 4" "masked_packed --model ViT-B-16 --masks union --stages 4 --masks-packed" \
           "masked_frombase --model ViT-B-16 --masks union --stages 4 --from-base fused" "frombase --from-base fused" \
           "n2 --tasks 2" "n4 --tasks 4" "n12 --tasks 12 --steps 10" "n16 --tasks 16 --steps 10" "n24 --tasks 24 --steps 8" "n32 --tasks 32 --steps 6"; do
  set -- $cfg; tag=$1; shift
  timeout -k 10 300 python bench.py --no-cpu "$@" > $O/r03_line_$tag.log 2>&1 || { tail -n 5 $O/r03_line_$tag.log; exit 1; }
  python - $O/r03_line_$tag.log "$tag" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(f"[{sys.argv[2]}]", d["ms_per_step"], (d.get("untuned") or {}).get("ms_per_step"), d["kernels_ms"], "path", d.get("path_roofline_frac"))
PY
done
/opt/rocm/bin/hipcc -O3 -std=c++17 -Wno-unused-value --offload-arch=gfx950 -I svd-quantization-task-merging_amd/csrc tools/probe/eig_time.hip -o /tmp/eig_time && \
  { /tmp/eig_time 8 296; /tmp/eig_time 8 13; /tmp/eig_time 20 296; } > $O/r03_eig_time.txt 2>&1
timeout -k 10 200 python tools/shard_one.py > $O/r03_shard_one.txt 2>&1; tail -n 7 $O/r03_shard_one.txt
timeout -k 10 200 python tools/shard_one.py 20 > $O/r03_shard_one_n20.txt 2>&1; tail -n 3 $O/r03_shard_one_n20.txt
timeout -k 10 200 python tools/host_overhead.py > $O/r03_host_overhead.txt 2>&1; head -n 6 $O/r03_host_overhead.txt
timeout -k 10 200 python tools/bench_aux.py > $O/r03_aux_kernels.jsonl 2>&1; tail -n 3 $O/r03_aux_kernels.jsonl | cut -c1-200
