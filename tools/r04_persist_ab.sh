#!/bin/bash
# A/B of the persistent scheduling of the streaming passes against the round-3 scheduling (gpurun_ab/libsvdq_r04base.so)
out=gpurun_out/r04_persist_ab.txt
: > $out
for rep in 1 2; do
for lib in gpurun_ab/libsvdq_r04base.so svd-quantization-task-merging_amd/libsvdq_hip.so; do
  for c in "" "--shard-of 8" "--shard-of 4" "--model ViT-B-32" "--tasks 16" "--tasks 4"; do
    SVDQ_LIB_PATH=$PWD/$lib timeout -k 10 300 python bench.py $c --no-cpu --placement-candidates 1 --steps 30 --warmup 5 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib [$c]', d['ms_per_step'], d.get('kernels_ms'))" >> $out || exit 1
  done
done
done
