#!/bin/bash
# Are the run-to-run differences of the norm backward ("Reproducibility finding") stale SCALAR-cache reads
# of the per-slice partial sums?  Builds dcl_bn.hip with -DDCL_BN_PROBE=<bits> (1: partial sums through agent-scope vector
# loads, 2: per-channel statistics / affine parameters too) next to the product library and counts differing runs of the
# reproducer with each:        (build here)  bash tools/probes/bn_coherence.sh build
#                              (GPU box)     gpurun -- bash tools/probes/bn_coherence.sh run
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/../.. && pwd)}
PKG=$ROOT/eccv2022-multi-scale-and-cross-scale-contrastive-segmentation_amd
OUT=$ROOT/tools/probes/variants
VARIANTS="${VARIANTS:-1 3}"
RUNS=${RUNS:-32}
if [ "$1" = build ]; then
  mkdir -p $OUT
  for v in $VARIANTS; do
    ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DDCL_BN_PROBE=$v -c $PKG/csrc/dcl_bn.hip -o $OUT/bn_$v.o &&
      hipcc -shared -fPIC --offload-arch=gfx950 $(ls $PKG/csrc/build/*.o | grep -v dcl_bn.o) $OUT/bn_$v.o -o $OUT/libdcl_bnprobe_$v.so ) &
  done
  wait
  ls -la $OUT/*bnprobe*.so
else
  cd $ROOT
  for mode in "whole on interleave" "whole alt group"; do
    for rep in 1 2; do
      python3 tools/probes/dbg_merged.py $mode runs=$RUNS | tail -3
      for v in $VARIANTS; do
        DCL_LIB_PATH=$OUT/libdcl_bnprobe_$v.so python3 tools/probes/dbg_merged.py $mode runs=$RUNS | tail -3
      done
    done
  done
fi
