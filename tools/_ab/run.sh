set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_full_size.py -x -q -m gpu > gpurun_out/ab/tests.log 2>&1 || { tail -30 gpurun_out/ab/tests.log; exit 1; }
timeout -k 10 400 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "fused" > gpurun_out/ab/fuzz.log 2>&1 || { tail -30 gpurun_out/ab/fuzz.log; exit 1; }
PDX_LIB_PATH=$PWD/tools/_ab/libpdx_tm.so timeout -k 10 300 python tools/_ab/timing.py
for i in 1 2 3; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-check 2>/dev/null | tail -1 > gpurun_out/ab/new_$i.json
  PDX_LIB_PATH=$PWD/tools/_ab/libpdx_t0.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-check 2>/dev/null | tail -1 > gpurun_out/ab/old_$i.json
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/*.json')):
    j=json.loads(open(f).read()); print(f, round(j['ms_per_step'],3), j.get('roofline',{}).get('kernel_ms_per_step'))
P
tail -2 gpurun_out/ab/tests.log; tail -2 gpurun_out/ab/fuzz.log
