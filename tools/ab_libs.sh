#!/bin/bash
# A/B of prebuilt library variants on ONE box: usage tools/ab_libs.sh libA.so libB.so ... (files under gpurun_ab/);
# each variant is copied over the in-tree library and benched with every argument line of $CASES (separated by ';')
R=$PWD
cp $R/svd-quantization-task-merging_amd/libsvdq_hip.so /tmp/lib_keep.so
IFS=';' read -ra cases <<< "${CASES:---diagnostics --tasks 20}"
for rep in 1 2; do
for lib in "$@"; do
  cp $R/gpurun_ab/$lib $R/svd-quantization-task-merging_amd/libsvdq_hip.so
  for c in "${cases[@]}"; do
    ms=$(timeout -k 10 300 python3 $R/bench.py $c --steps 20 --warmup 3 --no-cpu 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('kernels_ms',''))")
    echo "$lib [$c] $ms" | tee -a $R/gpurun_out/ab_libs.txt
  done
done
done
cp /tmp/lib_keep.so $R/svd-quantization-task-merging_amd/libsvdq_hip.so
