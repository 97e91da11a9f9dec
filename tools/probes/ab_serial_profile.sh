#!/bin/bash
# Serialised kernel tables (every launch on one stream) of one steady-state step under two environments, for a per-kernel diff:
#   gpurun -- bash tools/probes/ab_serial_profile.sh <tag> "ENV_A" "ENV_B"   -> gpurun_out/<tag>_{a,b}_step_kernels_serial.csv
TAG=$1; A="$2"; B="$3"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for v in a b; do
  E="$A"; [ $v = b ] && E="$B"
  D=$OUT/${TAG}_${v}_trace; rm -rf $D; mkdir -p $D
  export DCL_BRANCH_STREAMS=0 DCL_HEAD_OVERLAP=0
  for kv in $E; do export $kv; done
  rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-eager-step > $D/run.log 2>&1 || tail -5 $D/run.log
  python3 $ROOT/tools/summarize_profile.py trace $(find $D -name '*kernel_trace.csv' | head -1) 3 5 > $OUT/${TAG}_${v}_step_kernels_serial.csv
  rm -rf $D
done
head -5 $OUT/${TAG}_a_step_kernels_serial.csv
