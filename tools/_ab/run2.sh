set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab2
for i in 1 2; do
for v in i10_10 i12_4 i12_2 i11_7; do
  PDX_LIB_PATH=$PWD/tools/_ab/libpdx_$v.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-check 2>/dev/null | tail -1 > gpurun_out/ab2/${v}_$i.json
done; done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab2/*.json')):
    j=json.loads(open(f).read()); print(f, round(j['ms_per_step'],3), j.get('roofline',{}).get('kernel_ms_per_step',{}).get('fused_last_digit_reduce'), j.get('checks'))
P
