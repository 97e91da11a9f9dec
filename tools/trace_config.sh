#!/bin/bash
# Kernel trace of a --config N training step, raw rows kept:  gpurun --timeout 900 -- bash tools/trace_config.sh 5
# -> gpurun_out/config${N}_kernel_trace.csv (all launches) and config${N}_kernels.csv (one steady-state step, summarised)
C=${1:-5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
D=/tmp/trace_cfg; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $ROOT/bench.py --config $C --steps 4 --warmup 3 --no-cpu-baseline --no-eager-step > $D/run.log 2>&1 || tail -5 $D/run.log
T=$(find $D -name '*kernel_trace.csv' | head -1)
python3 $ROOT/tools/summarize_profile.py trace $T 3 5 > $OUT/config${C}_kernels.csv
cp $T $OUT/config${C}_kernel_trace.csv
tail -2 $D/run.log | cut -c1-400
