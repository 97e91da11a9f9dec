#!/bin/bash
# Counter pass (rocprofv3 --pmc, own run, kernel trace only) over an arbitrary python tool, summarised per kernel:
#   gpurun -- bash tools/pmc_cmd.sh "<counters>" <grep pattern> tools/x.py args...
PMC=$1; PAT=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/pmc_cmd
rm -rf $OUT; mkdir -p $OUT
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT -- python3 $SCRIPT "$@" > $OUT/run.log 2>&1 || tail -5 $OUT/run.log
python3 $ROOT/tools/summarize_profile.py pmc $(find $OUT -name '*counter_collection.csv' | head -1) | grep -E "$PAT" | python3 -c "
import csv, sys
for r in csv.reader(sys.stdin):
    print(','.join([r[0][:72].replace(',', ';')] + r[1:]))"
