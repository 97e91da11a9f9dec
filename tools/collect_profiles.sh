#!/bin/bash
# Collects every rocprofv3 summary behind the numbers in DESIGN.md / bench.py for one round (run on the GPU box):
#   gpurun --timeout 2400 -- bash tools/collect_profiles.sh r02
# Outputs (gpurun_out/, then copied into profiles/ by hand):
#   <tag>_step_kernels.csv / _step_kernel_stats.csv   one steady-state training step (kernel trace)
#   <tag>_loss_kernel_stats.csv                       loss workload, rocprofv3's own per-kernel stats
#   <tag>_loss_pmc_{fetch,write,sq}.csv               FETCH_SIZE / WRITE_SIZE / SQ counters, separate passes
#   <tag>_conv_pmc_{fetch,write}.csv                  same for the direct convolution kernels (per-shape tool)
#   <tag>_conv_per_shape.csv                          HIP-event time per launch and roofline fraction per shape
#   <tag>_conv_per_shape_rocprof.csv                  rocprofv3 kernel trace of the same tool, one row per (kernel, grid)
#   <tag>_gemm_shapes_{swinl,swint,head}.csv          dcl_gemm_f16x3 vs the library's fp32 GEMM per shape (time, TFLOP/s, errors)
#   <tag>_gemm_pmc_{sq,fetch,write}.csv               counters of the GEMM on the Swin-L stage-3 fc1 shape
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
S="python3 $ROOT/tools/summarize_profile.py"

bash $ROOT/tools/profile_step.sh $TAG > /dev/null

D=$OUT/${TAG}_loss_trace; rm -rf $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $ROOT/bench.py --workload loss --steps 10 --warmup 3 --no-cpu-baseline > $D.log 2>&1
cp $(find $D -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_loss_kernel_stats.csv; rm -rf $D

for C in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $C | tr A-Z a-z | sed 's/_size//')
  D=$OUT/${TAG}_pmc_tmp; rm -rf $D
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $ROOT/bench.py --workload loss --steps 2 --warmup 1 --no-cpu-baseline > $D.log 2>&1
  $S pmc $(find $D -name '*counter_collection.csv' | head -1) > $OUT/${TAG}_loss_pmc_$n.csv; rm -rf $D
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $ROOT/tools/per_shape_roofline.py > $D.log 2>&1
  $S pmc $(find $D -name '*counter_collection.csv' | head -1) > $OUT/${TAG}_conv_pmc_$n.csv; rm -rf $D
done
D=$OUT/${TAG}_pmc_tmp; rm -rf $D
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d $D -- python3 $ROOT/bench.py --workload loss --steps 2 --warmup 1 --no-cpu-baseline > $D.log 2>&1
$S pmc $(find $D -name '*counter_collection.csv' | head -1) > $OUT/${TAG}_loss_pmc_sq.csv; rm -rf $D $OUT/*.log

python3 $ROOT/tools/per_shape_roofline.py --out $OUT/${TAG}_conv_per_shape.csv > /dev/null
# the same tool under the kernel trace: one rocprof row per (kernel, grid) = per shape (the head launch among them)
D=$OUT/${TAG}_shape_trace; rm -rf $D
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $ROOT/tools/per_shape_roofline.py > $D.log 2>&1
$S bygrid $(find $D -name '*kernel_trace.csv' | head -1) > $OUT/${TAG}_conv_per_shape_rocprof.csv; rm -rf $D $D.log
# split-f16 GEMM: per-shape table against the library, counters of the Swin-L stage-3 fc1 shape
python3 $ROOT/tools/gemm_shapes.py --only swinl > $OUT/${TAG}_gemm_shapes_swinl.csv 2>/dev/null
python3 $ROOT/tools/gemm_shapes.py --only swint > $OUT/${TAG}_gemm_shapes_swint.csv 2>/dev/null
python3 $ROOT/tools/gemm_shapes.py --only head > $OUT/${TAG}_gemm_shapes_head.csv 2>/dev/null
D=$OUT/${TAG}_pmc_tmp; rm -rf $D
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d $D -- python3 $ROOT/tools/gemm_shapes.py --shape 25600,768,3072 --no-library --iters 3 > $D.log 2>&1
$S pmc $(find $D -name '*counter_collection.csv' | head -1) | grep -v "at::native" > $OUT/${TAG}_gemm_pmc_sq.csv; rm -rf $D
for C in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $C | tr A-Z a-z | sed 's/_size//')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $ROOT/tools/gemm_shapes.py --shape 25600,768,3072 --no-library --iters 3 > $D.log 2>&1
  $S pmc $(find $D -name '*counter_collection.csv' | head -1) | grep -v "at::native" > $OUT/${TAG}_gemm_pmc_$n.csv; rm -rf $D
done
rm -f $OUT/*.log
# configs 4 / 5: one steady-state step each (the aggregate of a run also holds MIOpen's solver search of the first step)
cd $ROOT && bash tools/profile_config4.sh $TAG 4 > /dev/null 2>&1
cd $ROOT && bash tools/profile_config4.sh $TAG 5 > /dev/null 2>&1; cd /tmp
ls -la $OUT | grep ${TAG}_
