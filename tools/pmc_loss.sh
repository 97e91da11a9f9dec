#!/bin/bash
# SQ counter pass over the loss workload (one rocprofv3 run, counters only + kernel trace), summarised per kernel.
#   gpurun -- bash tools/pmc_loss.sh <tag>
set -e
TAG=${1:-pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${TAG}_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc ${PMC:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE} \
    --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --workload loss --steps 2 --warmup 1 --no-cpu-baseline > $OUT/run.log 2>&1 || tail -5 $OUT/run.log
F=$(find $OUT -name '*counter_collection.csv' | head -1)
python3 $ROOT/tools/summarize_profile.py pmc $F > $ROOT/gpurun_out/${TAG}_sq.csv
grep -E "k_sweep" $ROOT/gpurun_out/${TAG}_sq.csv
