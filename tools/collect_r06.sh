#!/bin/bash
# Round-6 profile collection (run on the GPU box):  gpurun --timeout 2400 -- bash tools/collect_r06.sh
# Outputs in gpurun_out/ (copied into profiles/ by hand):
#   r06_step_kernels.csv, r06_step_kernel_stats.csv, r06_queue_busy.txt, r06_step_timeline.txt   one steady-state step, multi-stream
#   r06_step_kernels_serial.csv       the same step with every launch on ONE stream (what bench.py's kernel table measures)
#   r06_loss_kernel_stats.csv         loss workload, rocprofv3 --stats
#   r06_loss_pmc_{fetch,write}.csv    FETCH_SIZE / WRITE_SIZE per kernel, separate passes (stream-K with 4 column slices)
#   r06_loss_pmc_{fetch,write}_1slice.csv   the same with one slice (round 3's partition)
#   r06_loss_pmc_fetch_nhwc.csv       K3 / K6 on channels-last embedding maps (the projector's training output)
#   r06_step_pmc_{fetch,write}.csv    FETCH_SIZE / WRITE_SIZE per kernel symbol over training steps
#   r06_conv_per_shape.csv            HIP-event time per launch and roofline fraction per (kernel, shape)
set -e
TAG=r06
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
S="python3 $ROOT/tools/summarize_profile.py"

bash $ROOT/tools/profile_step.sh $TAG > /dev/null

D=$OUT/${TAG}_serial_trace; rm -rf $D; mkdir -p $D
DCL_BRANCH_STREAMS=0 DCL_HEAD_OVERLAP=0 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $ROOT/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-eager-step > $D/run.log 2>&1 || tail -5 $D/run.log
$S trace $(find $D -name '*kernel_trace.csv' | head -1) 3 5 > $OUT/${TAG}_step_kernels_serial.csv; rm -rf $D

D=$OUT/${TAG}_loss_trace; rm -rf $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $ROOT/bench.py --workload loss --steps 10 --warmup 3 --no-cpu-baseline > $D.log 2>&1
cp $(find $D -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_loss_kernel_stats.csv; rm -rf $D

for C in FETCH_SIZE WRITE_SIZE; do
  n=$(echo $C | tr A-Z a-z | sed 's/_size//')
  D=$OUT/${TAG}_pmc_tmp; rm -rf $D
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $ROOT/bench.py --workload loss --steps 2 --warmup 1 --no-cpu-baseline > $D.log 2>&1
  $S pmc $(find $D -name '*counter_collection.csv' | head -1) > $OUT/${TAG}_loss_pmc_$n.csv; rm -rf $D
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-eager-step > $D.log 2>&1
  $S pmc $(find $D -name '*counter_collection.csv' | head -1) > $OUT/${TAG}_step_pmc_$n.csv; rm -rf $D
done
rm -f $OUT/*.log
python3 $ROOT/tools/per_shape_roofline.py --out $OUT/${TAG}_conv_per_shape.csv > /dev/null 2>&1 || true
ls -la $OUT | grep ${TAG}_
# round 6: the deferred-norm pieces against what they replace (stand-alone), and the step boundary without a profiler
python3 $ROOT/tools/probes/conv_pre_time.py 2>/dev/null | grep " ch " > $OUT/${TAG}_conv_pre_time.txt || true
python3 $ROOT/tools/step_boundary.py --steps 20 --warmup 3 2>/dev/null > $OUT/${TAG}_step_boundary.txt || true
ls -la $OUT | grep ${TAG}_
