#!/bin/bash
# Round-4 profile collection for the compression path on the GPU box (run from the repo root); the consumer legs are
# tools/profile_r04_consumers.sh.  Every rocprofv3 invocation runs once; counter passes use --pmc with --kernel-trace
# only.  Outputs land in gpurun_out/ (r04_*), the ones DESIGN.md quotes are copied to profiles/.
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out
# 1. the default bench line (with the CPU baseline), as the driver runs it
timeout -k 10 400 python bench.py > $O/r04_bench_default.log 2>&1 || { tail -n 5 $O/r04_bench_default.log; exit 1; }
grep '^{' $O/r04_bench_default.log > $O/r04_bench_line.json; cut -c1-260 $O/r04_bench_line.json
# 2. kernel statistics + timed-region averages: headline, masked config #3, N = 20
bash tools/r04_prof1.sh bench || exit 1
bash tools/r04_prof1.sh masked_vitb16 --model ViT-B-16 --masks union --stages 4 || exit 1
bash tools/r04_prof1.sh n20 --tasks 20 || exit 1
# 3. HBM traffic (FETCH_SIZE, WRITE_SIZE), one pass each
bash tools/r04_pmc_traffic.sh n8 && bash tools/r04_pmc_traffic.sh masked_vitb16 --model ViT-B-16 --masks union --stages 4 && \
  bash tools/r04_pmc_traffic.sh n20 --tasks 20 || exit 1
# 4. other bench lines
: > $O/r04_other_configs.jsonl
for cfg in "b32 --model ViT-B-32" "b16 --model ViT-B-16 --stages 4" "masked_vitb16 --model ViT-B-16 --masks union --stages 4" \
           "masked_packed --model ViT-B-16 --masks union --stages 4 --masks-packed" "frombase --from-base fused" \
           "n2 --tasks 2" "n4 --tasks 4" "n12 --tasks 12 --steps 10" "n16 --tasks 16 --steps 10" "n20 --tasks 20 --steps 10" \
           "n24 --tasks 24 --steps 8" "n32 --tasks 32 --steps 6"; do
  set -- $cfg; tag=$1; shift
  timeout -k 10 300 python bench.py --no-cpu "$@" 2>/dev/null | grep '^{' >> $O/r04_other_configs.jsonl || { echo "$tag failed"; exit 1; }
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_other_configs.jsonl"):
    d = json.loads(l); print(d["config"]["workload"][:70], d["ms_per_step"], (d.get("untuned") or {}).get("ms_per_step"), d["kernels_ms"], "path", d.get("path_roofline_frac"))
PY
timeout -k 10 200 python tools/shard_one.py > $O/r04_shard_one.txt 2>&1; tail -n 7 $O/r04_shard_one.txt
timeout -k 10 200 python tools/host_overhead.py > $O/r04_host_overhead.txt 2>&1; head -n 6 $O/r04_host_overhead.txt
timeout -k 10 200 python tools/bench_aux.py > $O/r04_aux_kernels.jsonl 2>&1; tail -n 3 $O/r04_aux_kernels.jsonl | cut -c1-200
