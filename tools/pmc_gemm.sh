#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of the GEMM on the roofline shapes of bench.py --config 4 | 5 (separate --pmc passes):
#   gpurun -- bash tools/pmc_gemm.sh   ->  gpurun_out/r04_gemm_pmc.csv
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "shape,counter,kernel,launches,avg_KiB_per_launch,max_KiB" > $OUT/r04_gemm_pmc.csv
for SH in "16384 384 1536" "25600 768 3072"; do
  for C in FETCH_SIZE WRITE_SIZE; do
    D=/tmp/pmc_gemm; rm -rf $D
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $ROOT/tools/probes/gemm_one.py $SH > $D.log 2>&1
    python3 $ROOT/tools/summarize_profile.py pmc $(find $D -name '*counter_collection.csv' | head -1) | grep k_gemm | sed "s/^/\"$SH\",/" >> $OUT/r04_gemm_pmc.csv
  done
done
cat $OUT/r04_gemm_pmc.csv
