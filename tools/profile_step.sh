#!/bin/bash
# Kernel trace of the training-step benchmark and the per-kernel table of ONE steady-state step.
#   gpurun -- bash tools/profile_step.sh <tag>      -> gpurun_out/<tag>_step_kernels.csv (+ rocprofv3's own stats)
set -e
TAG=${1:-step}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-eager-step > $OUT/run.log 2>&1 || tail -5 $OUT/run.log
F=$(find $OUT -name '*kernel_trace.csv' | head -1)
python3 $ROOT/tools/summarize_profile.py trace $F 3 5 > $ROOT/gpurun_out/${TAG}_step_kernels.csv
cp $(find $OUT -name '*kernel_stats.csv' | head -1) $ROOT/gpurun_out/${TAG}_step_kernel_stats.csv
python3 $ROOT/tools/queue_busy.py $F 5 > $ROOT/gpurun_out/${TAG}_queue_busy.txt 2>&1 || true
python3 $ROOT/tools/step_timeline.py $F 5 > $ROOT/gpurun_out/${TAG}_step_timeline.txt 2>&1 || true
head -40 $ROOT/gpurun_out/${TAG}_step_kernels.csv
# the raw trace is large: keep only the summaries
rm -rf $OUT
