#!/bin/bash
for mb in 0 64 100 200 400; do for ur in 1024 2048 4096; do
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu --unit-rows $ur --pipeline-mb $mb 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('pipeline_mb $mb unit_rows $ur groups', d['config']['pipeline_groups'], 'ms', d['ms_per_step'], 'frac', d['path_roofline_frac'])"
done; done
