#!/bin/bash
# kernel stats of a bench configuration under rocprofv3 (csv), averaged over the timed steps
# usage: bash tools/r04_prof1.sh <tag> <bench args...>
set -o pipefail
tag=$1; shift
mkdir -p gpurun_out/r04_prof_$tag
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_prof_$tag -o t -- python3 bench.py --no-cpu --steps 10 --warmup 2 "$@" > gpurun_out/r04_prof_$tag.log 2>&1
rc=$?; grep '^{' gpurun_out/r04_prof_$tag.log | cut -c1-250; [ $rc -ne 0 ] && { tail -n 5 gpurun_out/r04_prof_$tag.log; exit $rc; }
f=$(find gpurun_out/r04_prof_$tag -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timed_region.py "$f" 10 gpurun_out/r04_${tag}_timed_region.json
cp $(find gpurun_out/r04_prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/r04_${tag}_kernel_stats.csv
rm -rf gpurun_out/r04_prof_$tag
