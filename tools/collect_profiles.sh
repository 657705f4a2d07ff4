#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through gpurun) and writes summaries under gpurun_out/;
# tools/summarize_profiles.py then condenses them into profiles/ (tracked).  Counters are collected in their own passes
# (--pmc with --kernel-trace only), FETCH_SIZE and WRITE_SIZE separately (TCC slots), as MI355X_MICROARCH.md prescribes.
set -e
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
# headline step alone (what `value` / `roofline` are computed from), then the default run with the secondary configs
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/kt.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_secondary" -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/kt_secondary.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-check --no-secondary > "$OUT/pmc_$c.log" 2>&1
done
# the same three collections for the general-keys plan (every key through the hash table): its own kernel trace and its own traffic
export PDX_GROUPBY_DENSE=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_general" -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/kt_general.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_general_$c" -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-check --no-secondary > "$OUT/pmc_general_$c.log" 2>&1
done
unset PDX_GROUPBY_DENSE
python bench.py --steps 5 --warmup 2 > "$OUT/bench.json" 2> "$OUT/bench.err"
PDX_GROUPBY_DENSE=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_general_keys.json" 2> "$OUT/bench_general_keys.err"
tail -1 "$OUT/bench.json"
