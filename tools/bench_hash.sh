#!/bin/bash
# general-keys step of bench.py (every key through the hash table): ms per step + per-kernel ms, for A/B of hash-path changes
export PDX_GROUPBY_DENSE=0
timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/b_hash.json 2> gpurun_out/b_hash.err && python -c "
import json; d=json.loads(open('gpurun_out/b_hash.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
