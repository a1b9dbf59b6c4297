#!/bin/bash
# The consumer legs on the round-3 binary: bench.py --merge / --diagnostics over the BASELINE shapes, one JSON line each
# (gpurun_out/r03_consumers.jsonl), then rocprofv3 kernel statistics of the two masked legs.
R=$PWD
mkdir -p $R/gpurun_out
out=$R/gpurun_out/r03_consumers.jsonl
: > $out
run() { timeout -k 10 300 python3 $R/bench.py "$@" --steps 20 2>/dev/null | tail -1 >> $out || exit 1; }
run --merge
run --merge --model ViT-B-16 --stages 4
run --merge --model ViT-B-16 --stages 4 --masks union
run --merge --tasks 20 --clusters 2 --no-cpu
run --diagnostics
run --diagnostics --model ViT-B-16 --stages 4
run --diagnostics --model ViT-B-16 --stages 4 --masks union
run --diagnostics --tasks 16
run --diagnostics --tasks 20
python3 - <<PY
import json
for l in open("$out"):
    d = json.loads(l)
    print(d["config"]["workload"][:60], d["config"].get("masks") and "masked" or "", d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
PY
cd /tmp && export TMPDIR=/tmp
for leg in merge diagnostics; do
  rm -rf /tmp/cons_$leg
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cons_$leg -o p -- python3 $R/bench.py --$leg --model ViT-B-16 --stages 4 --masks union --steps 20 --no-cpu > /tmp/cons_$leg.log 2>&1 < /dev/null
  f=$(find /tmp/cons_$leg -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -12 "$f" > $R/gpurun_out/r03_masked_${leg}_kernel_stats.csv
done
head -6 $R/gpurun_out/r03_masked_merge_kernel_stats.csv | cut -c1-200
head -6 $R/gpurun_out/r03_masked_diagnostics_kernel_stats.csv | cut -c1-200
