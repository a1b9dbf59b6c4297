#!/bin/bash
# A/B of build variants of the streaming kernels (SVDQ_LIB_PATH selects the library); prints ms/step and kernels.
# Variants are built by hand, e.g.:  hipcc ... -DSVDQ_UNROLL_BP2=2 -c svdq_project.hip -o build/project_a.o; link as var_a.so
for cfg in ${CFGS:-ViT-L-14:20 ViT-L-14:32}; do m=${cfg%%:*}; n=${cfg##*:}
for v in libsvdq_hip var_a var_b var_c; do
  [ -f svd-quantization-task-merging_amd/$v.so ] || continue
  SVDQ_LIB_PATH=$PWD/svd-quantization-task-merging_amd/$v.so timeout -k 10 250 python bench.py --steps 6 --warmup 2 --no-cpu --placement-candidates 1 --model $m --tasks $n 2>/dev/null < /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$m x $n $v', d['ms_per_step'], d['kernels_ms'])"
done; done
