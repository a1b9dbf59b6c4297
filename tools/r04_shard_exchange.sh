#!/bin/bash
# What one GPU of the 8-GPU strong-scaling run does per step, on one GPU: rank 0's LPT share, without and with the
# RCCL exchange (a one-rank world run as a distributed job).  Output: gpurun_out/r04_shard_exchange.txt
out=gpurun_out/r04_shard_exchange.txt
: > $out
for k in 8 4 2; do
  echo "== --shard-of $k, no process group" >> $out
  python bench.py --shard-of $k --no-cpu --placement-candidates 1 --steps 50 --warmup 5 2>>$out | cut -c1-400 >> $out || exit 1
  echo "== --shard-of $k, one-rank world over RCCL (planned gather, overlap)" >> $out
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 SVDQ_DIST_SINGLE=1 \
    python bench.py --shard-of $k --no-cpu --placement-candidates 1 --steps 50 --warmup 5 2>>$out | cut -c1-400 >> $out || exit 1
done
