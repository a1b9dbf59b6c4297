#!/bin/bash
# A/B of the persistent scheduling of the streaming passes (experiment; gpurun_ab/libsvdq_persist.so built from
# profiles/experiments/r04_persistent_scheduling.patch, SVDQ_PERSIST = bit 0 pass 1 | bit 1 pass 2) against the round-3
# scheduling (gpurun_ab/libsvdq_r04base.so), interleaved repetitions on one box
out=gpurun_out/r04_persist_ab2.txt
: > $out
for rep in 1 2 3 4 5; do
for v in "r04base 0" "persist 3" "persist 2" "persist 1"; do
  set -- $v
  for c in "" "--shard-of 8"; do
    SVDQ_PERSIST=$2 SVDQ_LIB_PATH=$PWD/gpurun_ab/libsvdq_$1.so timeout -k 10 300 python bench.py $c --no-cpu --placement-candidates 1 --steps 30 --warmup 5 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 SVDQ_PERSIST=$2 [$c]', d['ms_per_step'], d.get('kernels_ms'))" >> $out || exit 1
  done
done
done
