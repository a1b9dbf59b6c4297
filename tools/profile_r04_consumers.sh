#!/bin/bash
# Round-4 evidence for the consumer kernels (run from the repo root on the GPU box): bench lines, rocprofv3 kernel
# statistics + timed-region averages, HBM traffic and SQ counter passes of bench.py --diagnostics / --merge.
# Every rocprofv3 invocation runs once; counter passes use --pmc with --kernel-trace only.  Outputs: gpurun_out/r04_*.
set -o pipefail
O=gpurun_out; mkdir -p $O
: > $O/r04_consumers.jsonl
run() { timeout -k 10 300 python3 bench.py "$@" --steps 20 2>/dev/null | tail -1 >> $O/r04_consumers.jsonl || exit 1; }
run --diagnostics
run --diagnostics --tasks 3 --no-cpu
run --diagnostics --tasks 12 --no-cpu
run --diagnostics --tasks 16 --no-cpu
run --diagnostics --tasks 20 --no-cpu
run --diagnostics --tasks 24 --no-cpu
run --diagnostics --tasks 32 --no-cpu
run --diagnostics --model ViT-B-16 --stages 4
run --diagnostics --model ViT-B-16 --stages 4 --masks union
run --merge
run --merge --tasks 16 --no-cpu
run --merge --tasks 20 --clusters 2 --no-cpu
run --merge --model ViT-B-16 --stages 4
run --merge --model ViT-B-16 --stages 4 --masks union
python3 - <<PY
import json
for l in open("$O/r04_consumers.jsonl"):
    d = json.loads(l)
    print(d["config"]["workload"][:58], "masked" if "mask" in json.dumps(d["config"]) and "union" in json.dumps(d["config"]) else "", d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
PY
bash tools/r04_prof1.sh diag --diagnostics || exit 1
bash tools/r04_prof1.sh diag_n20 --diagnostics --tasks 20 || exit 1
bash tools/r04_prof1.sh merge --merge || exit 1
bash tools/r04_prof1.sh merge_n20c2 --merge --tasks 20 --clusters 2 || exit 1
bash tools/r04_prof1.sh diag_masked_vitb16 --diagnostics --model ViT-B-16 --stages 4 --masks union || exit 1
bash tools/r04_prof1.sh merge_masked_vitb16 --merge --model ViT-B-16 --stages 4 --masks union || exit 1
bash tools/r04_pmc_sq.sh diag --diagnostics > /dev/null
bash tools/r04_pmc_sq.sh diag_n20 --diagnostics --tasks 20 > /dev/null
bash tools/r04_pmc_sq.sh merge --merge > /dev/null
bash tools/r04_pmc_sq.sh merge_n20c2 --merge --tasks 20 --clusters 2 > /dev/null
bash tools/r04_pmc_traffic.sh diag --diagnostics
bash tools/r04_pmc_traffic.sh merge --merge
grep -h "k_diag<\|k_merge_reconstruct<" $O/r04_pmc_sq_diag.txt $O/r04_pmc_sq_merge.txt | cut -c1-400
