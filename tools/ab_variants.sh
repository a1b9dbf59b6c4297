#!/bin/bash
timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('baseline ms', d['ms_per_step'], d['kernels_ms'], 'frac', d['path_roofline_frac'])"
for cfg in "64 2 2048" "64 3 2048" "64 4 2048" "64 3 4096" "64 3 1024" "128 3 2048" "256 3 2048"; do set -- $cfg
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu --unit-rows $3 --pipeline-mb $1 --pipeline-lag $2 --pipeline-in-c 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('C-pipeline mb $1 lag $2 unit_rows $3 ms', d['ms_per_step'], 'frac', d['path_roofline_frac'])"
done
