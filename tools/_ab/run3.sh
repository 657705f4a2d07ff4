set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ab3
for v in 3 6 12 24 48 64; do
  PDX_FLR_WGS_PER_CU=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-check 2>/dev/null | tail -1 > gpurun_out/ab3/w$v.json
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab3/*.json')):
    j=json.loads(open(f).read()); print(f, round(j['ms_per_step'],3), j.get('roofline',{}).get('kernel_ms_per_step',{}).get('fused_last_digit_reduce'))
P
