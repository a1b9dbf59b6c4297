#!/bin/bash
for round in 1 2; do
for v in libsvdq_hip var_ntl var_nts var_ntls; do
  SVDQ_LIB_PATH=$PWD/svd-quantization-task-merging_amd/$v.so timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['kernels_ms'])"
done; done
