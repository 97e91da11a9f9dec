mkdir -p gpurun_out/r4n
B="python bench.py --steps 8 --warmup 3 --no-eager-step --no-cpu-baseline --no-reference-config"
for i in 1 2; do
  for c in 5 4; do
    DCL_FUSED_MLP=0 $B --config $c > gpurun_out/r4n/c${c}_off_$i.json 2> gpurun_out/r4n/err.log || tail -3 gpurun_out/r4n/err.log
    DCL_FUSED_MLP=1 $B --config $c > gpurun_out/r4n/c${c}_on_$i.json 2> gpurun_out/r4n/err.log || tail -3 gpurun_out/r4n/err.log
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4n/c*_o*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['value'])
    except Exception as e: print(f, 'ERR', e)
PY
