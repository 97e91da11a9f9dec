#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python tool:  gpurun -- bash tools/prof_cmd.sh <n rows> tools/x.py args...
N=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/prof_cmd
rm -rf $OUT; mkdir -p $OUT
SCRIPT=$ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $SCRIPT "$@" > $OUT/run.log 2>&1
grep -v "^W2026\|amdgpu.ids" $OUT/run.log | tail -40
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:$N]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1e3:9.1f} us  min {float(r['MinNs']) / 1e3:9.1f}")
PY
