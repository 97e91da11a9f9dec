ROOT=$(pwd)
OUT=$ROOT/gpurun_out/lh_trace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --config 4 --steps 4 --warmup 3 --no-cpu-baseline --no-eager-step > $OUT/run.log 2>&1
F=$(find $OUT -name '*kernel_trace.csv' | head -1)
python3 - "$F" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t=[int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"].startswith("k_label_hist")]
print(len(t)); print([round((b-a)/1e6,2) for a,b in zip(t,t[1:])])
PY
rm -rf $OUT
