#!/bin/bash
# Kernel trace of the config-4 benchmark (UPerNet + Swin-T) and the per-kernel table of ONE steady-state step.
#   gpurun -- bash tools/profile_config4.sh <tag> [config]      -> gpurun_out/<tag>_config<config>_kernels.csv  (config 4 | 5)
set -e
TAG=${1:-c4}
CFG=${2:-4}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_c4_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --config $CFG --steps 4 --warmup 3 --no-cpu-baseline --no-eager-step > $OUT/run.log 2>&1 || tail -5 $OUT/run.log
F=$(find $OUT -name '*kernel_trace.csv' | head -1)
python3 $ROOT/tools/summarize_profile.py trace $F 4 5 > $ROOT/gpurun_out/${TAG}_config${CFG}_kernels.csv
head -50 $ROOT/gpurun_out/${TAG}_config${CFG}_kernels.csv
rm -rf $OUT
