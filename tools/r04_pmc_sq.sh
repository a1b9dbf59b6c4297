#!/bin/bash
# SQ counters of one bench configuration, two passes of <= 8 counters, each pass run ONCE (rocprofv3 --pmc with
# --kernel-trace only).  usage: bash tools/r04_pmc_sq.sh <tag> <bench args...>; output gpurun_out/r04_pmc_sq_<tag>.txt
tag=$1; shift
R=$PWD
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/r04_pmc_sq_$tag.txt
for pass in "a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA"; do
  set -- $pass "$@"; p=$1; shift; ctrs=(); while [ "${1#SQ_}" != "$1" ]; do ctrs+=("$1"); shift; done
  rm -rf /tmp/pmc_$p
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "${ctrs[@]}" --output-format csv -d /tmp/pmc_$p -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --placement-candidates 1 "$@" > /tmp/pmc_$p.log 2>&1 < /dev/null
  rc=$?
  f=$(find /tmp/pmc_$p -name "*counter_collection.csv" | head -1)
  echo "# pass $p: ${ctrs[*]} (exit $rc)" >> $R/gpurun_out/r04_pmc_sq_$tag.txt
  if [ -n "$f" ]; then python3 $R/tools/pmc_summary.py "$f" >> $R/gpurun_out/r04_pmc_sq_$tag.txt 2>&1; else tail -5 /tmp/pmc_$p.log >> $R/gpurun_out/r04_pmc_sq_$tag.txt; fi
done
cat $R/gpurun_out/r04_pmc_sq_$tag.txt | cut -c1-420
