#!/bin/bash
# Everything the round's last commit is judged on, from ONE box:  gpurun --timeout 3000 -- bash tools/collect_r04_final.sh
#   tools/collect_r04.sh (step / loss traces, PMC passes, per-shape table), one steady-state step of configs 4 and 5, then the bench lines
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
bash tools/collect_r04.sh > $OUT/collect.log 2>&1 || tail -5 $OUT/collect.log
for C in 4 5; do
  bash tools/trace_config.sh $C > /dev/null 2>&1
  mv $OUT/config${C}_kernels.csv $OUT/r04_config${C}_kernels.csv; rm -f $OUT/config${C}_kernel_trace.csv
done
cd $ROOT
python3 bench.py --kernel-table $OUT/r04_kernel_table.json > $OUT/r04_bench_default.json 2> $OUT/bench_default.err
python3 bench.py --workload loss --steps 30 --warmup 5 > $OUT/r04_bench_loss.json 2> /dev/null
python3 bench.py --labels blocky --no-cpu-baseline > $OUT/r04_bench_blocky.json 2> /dev/null
python3 bench.py --config 4 --steps 8 --warmup 3 --no-cpu-baseline --no-eager-step > $OUT/r04_bench_config4.json 2> /dev/null
python3 bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline --no-eager-step > $OUT/r04_bench_config5.json 2> /dev/null
for f in default loss blocky config4 config5; do python3 -c "import json,sys; d=json.loads(open('$OUT/r04_bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'])"; done
