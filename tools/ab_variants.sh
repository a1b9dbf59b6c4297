#!/bin/bash
for round in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('ms', d['ms_per_step'], d['kernels_ms'], 'frac', d['path_roofline_frac'])"
done
