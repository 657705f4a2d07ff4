#!/bin/bash
# Counter passes over bench.py (one --pmc set per pass, kernel trace only), summarised per kernel by tools/pmc_table.py.
# usage: tools/pmc_passes.sh TAG "SET1 counters" "SET2 counters" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
OUT=gpurun_out/pmc_$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
i=0
for set in "$@"; do
  i=$((i+1))
  # PMC_PROG (optional): another python program of this repo instead of the bench, e.g. PMC_PROG="tools/bench_nullsum.py"
  timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/p$i" -- python ${PMC_PROG:-bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-check} > "$OUT/p$i.log" 2>&1
  rc=$?
  echo "pass $i ($set): rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
done
python tools/pmc_table.py "$OUT" > "$OUT/table.txt" 2>&1
cat "$OUT/table.txt"
