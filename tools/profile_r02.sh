#!/bin/bash
# Round-2 profile collection on the GPU box (run from the repo root): kernel-trace statistics of the default bench
# command, then separate FETCH_SIZE / WRITE_SIZE counter passes (never combined with other trace domains).
# Outputs land in gpurun_out/; the ones to be judged are copied into profiles/ afterwards.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
EXTRA="$@"
tag=${TAG:-r02}
rm -rf /tmp/prof_$tag /tmp/pmc_f_$tag /tmp/pmc_w_$tag
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu $EXTRA > $R/gpurun_out/${tag}_prof_bench.log 2>&1 < /dev/null || { tail -n 5 $R/gpurun_out/${tag}_prof_bench.log; exit 1; }
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${tag}_bench_kernel_stats.csv && head -n 12 $R/gpurun_out/${tag}_bench_kernel_stats.csv | cut -c1-200
grep '^{' $R/gpurun_out/${tag}_prof_bench.log > $R/gpurun_out/${tag}_bench_line.json
kt=$(find /tmp/prof_$tag -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_timed_region.py "$kt" 10 $R/gpurun_out/${tag}_bench_timed_region.json
timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f_$tag -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --placement-candidates 1 $EXTRA > $R/gpurun_out/${tag}_pmc_f.log 2>&1 < /dev/null || { tail -n 5 $R/gpurun_out/${tag}_pmc_f.log; exit 1; }
timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w_$tag -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --placement-candidates 1 $EXTRA > $R/gpurun_out/${tag}_pmc_w.log 2>&1 < /dev/null || { tail -n 5 $R/gpurun_out/${tag}_pmc_w.log; exit 1; }
ff=$(find /tmp/pmc_f_$tag -name "*counter_collection.csv" | head -1)
fw=$(find /tmp/pmc_w_$tag -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_traffic.py "$ff" "$fw" $R/gpurun_out/${tag}_pmc_traffic.json
