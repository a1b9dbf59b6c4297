#!/bin/bash
# A/B of build variants of the streaming kernels (SVDQ_LIB_PATH selects the library); prints ms/step and kernels
for cfg in "ViT-L-14 20" "ViT-L-14 32" "ViT-L-14 16" "ViT-L-14 8"; do set -- $cfg
for v in libsvdq_hip var_a var_b var_c; do
  SVDQ_LIB_PATH=$PWD/svd-quantization-task-merging_amd/$v.so timeout -k 10 250 python bench.py --steps 6 --warmup 2 --no-cpu --model $1 --tasks $2 2>/dev/null < /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1 x $2 $v', d['ms_per_step'], d['kernels_ms'])"
done; done
