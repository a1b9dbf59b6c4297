#!/bin/bash
# SQ counters of the headline kernels (ViT-L-14 x 8), two passes of 7-8 counters each; run from the repo root on the GPU box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA"; do
  set -- $pass; tag=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$tag -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu $EXTRA > /tmp/pmc_$tag.log 2>&1 < /dev/null
  f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_summary.py "$f" > $R/gpurun_out/pmc_${OUT:-n8}_$tag.txt 2>&1 < /dev/null; else tail -5 /tmp/pmc_$tag.log > $R/gpurun_out/pmc_${OUT:-n8}_$tag.txt; fi
  cat $R/gpurun_out/pmc_${OUT:-n8}_$tag.txt
done
