#!/bin/bash
# step time of the headline query at (rows, keys) for several PDX_FUSED_LAST_DIGIT_MIN_RUN: where the fused last digit starts to pay
for cfg in "1e9 1e7" "5e8 1e7" "2.5e8 1e7" "1e8 1e6" "5e7 1e6" "2e7 1e6" "1e8 1e7" "3e7 3e5"; do
  set -- $cfg
  for mr in 8192 1024 256; do
    PDX_FUSED_LAST_DIGIT_MIN_RUN=$mr timeout -k 10 300 python bench.py --rows $1 --keys $2 --no-cpu-baseline --no-secondary --no-check 2>gpurun_out/bs.err | tail -1 > gpurun_out/bs.json
    python -c "
import json; d=json.loads(open('gpurun_out/bs.json').read()); k=d['roofline']['kernel_ms_per_step']; print('rows=$1 keys=$2 min_run=$mr', round(d['ms_per_step'],3), 'fused' if 'fused_last_digit_reduce' in k else 'classic')"
  done
done
