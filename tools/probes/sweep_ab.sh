#!/bin/bash
# Variant builds of csrc/dcl_sweep.hip for tools/probes/sweep_ab.py (run HERE; the .so files travel with the snapshot):
#   bash tools/probes/sweep_ab.sh build TWO=-DDCL_SWEEP_TWO_CHAINS ...
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
PKG=$ROOT/eccv2022-multi-scale-and-cross-scale-contrastive-segmentation_amd
OUT=$ROOT/tools/probes/variants
mkdir -p $OUT
shift || true
OTHERS=$(ls $PKG/csrc/build/*.o | grep -v dcl_sweep.o)
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}; [ "$flags" = "$spec" ] && flags=""
  ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -c $PKG/csrc/dcl_sweep.hip -o /tmp/dcl_sweep_$name.o \
    && hipcc -shared -fPIC --offload-arch=gfx950 $OTHERS /tmp/dcl_sweep_$name.o -o $OUT/libdcl_$name.so && echo built $name ) &
done
wait
