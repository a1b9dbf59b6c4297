#!/bin/bash
# A/B of fused-schedule build variants (SVDQ_LIB_PATH selects the library); prints ms/step
MODEL=${1:-ViT-B-32}
run() { SVDQ_LIB_PATH=$PWD/svd-quantization-task-merging_amd/$1.so timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu --model $MODEL $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$MODEL $1 $2', d['ms_per_step'], d['kernels_ms'])" || echo "$1 $2 FAILED"; }
run libsvdq_hip ""
for lag in 1 32 64 128; do run var_c "--fused --fused-lag-mb $lag"; done
run libsvdq_hip "--fused --fused-lag-mb 64"
