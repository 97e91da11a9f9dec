# per-shape convolution tables + FETCH / WRITE counters (the conv part of tools/collect_profiles.sh)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out; TAG=r03
cd /tmp && export TMPDIR=/tmp
S="python3 $ROOT/tools/summarize_profile.py"
for C in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $C | tr A-Z a-z | sed 's/_size//')
  D=/tmp/pmc_tmp; rm -rf $D
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $ROOT/tools/per_shape_roofline.py > $D.log 2>&1
  $S pmc $(find $D -name '*counter_collection.csv' | head -1) > $OUT/${TAG}_conv_pmc_$n.csv; rm -rf $D
done
python3 $ROOT/tools/per_shape_roofline.py --out $OUT/${TAG}_conv_per_shape.csv > /dev/null
D=/tmp/shape_trace; rm -rf $D
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $ROOT/tools/per_shape_roofline.py > $D.log 2>&1
$S bygrid $(find $D -name '*kernel_trace.csv' | head -1) > $OUT/${TAG}_conv_per_shape_rocprof.csv; rm -rf $D $D.log
grep "wgrad" $OUT/${TAG}_conv_per_shape.csv | cut -c1-120
