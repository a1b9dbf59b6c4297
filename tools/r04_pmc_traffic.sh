#!/bin/bash
# HBM traffic per kernel: one FETCH_SIZE pass and one WRITE_SIZE pass (rocprofv3 --pmc with --kernel-trace only), each
# run ONCE.  usage: bash tools/r04_pmc_traffic.sh <tag> <bench args...>; output gpurun_out/r04_pmc_traffic_<tag>.json
tag=$1; shift
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmct_$c
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmct_$c -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --placement-candidates 1 "$@" > /tmp/pmct_$c.log 2>&1 < /dev/null || { echo "$c pass failed"; tail -5 /tmp/pmct_$c.log; exit 1; }
done
f=$(find /tmp/pmct_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find /tmp/pmct_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_traffic.py "$f" "$w" $R/gpurun_out/r04_pmc_traffic_$tag.json "$@" | grep -v k_probe
