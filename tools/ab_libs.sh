#!/bin/bash
# A/B of prebuilt library variants: usage tools/ab_libs.sh "<bench args>" libA.so libB.so ... (files under gpurun_ab/);
# each variant is copied over the in-tree library and benched with every --tasks value in $TASKS
R=$PWD; args=$1; shift
cp $R/svd-quantization-task-merging_amd/libsvdq_hip.so /tmp/lib_keep.so
for lib in "$@"; do
  cp $R/gpurun_ab/$lib $R/svd-quantization-task-merging_amd/libsvdq_hip.so
  for n in ${TASKS:-20}; do
    ms=$(timeout -k 10 300 python3 $R/bench.py $args --tasks $n --steps 10 --warmup 3 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$lib tasks=$n ms=$ms" | tee -a $R/gpurun_out/ab_libs.txt
  done
done
cp /tmp/lib_keep.so $R/svd-quantization-task-merging_amd/libsvdq_hip.so
