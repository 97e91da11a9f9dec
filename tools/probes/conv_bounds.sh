#!/bin/bash
# Which part of a direct-convolution launch bounds it?  Builds probe variants of libdcl_hip.so (dcl_conv3x3.hip with
# -DDCL_CONV_PROBE=<bits>: 1 no MFMAs, 2 no patch loads, 4 no output stores, 8 no weight loads; results wrong) next to
# the product library and times the BasicBlock shapes with each:   (build here)  bash tools/probes/conv_bounds.sh build
#                                                                  (GPU box)     gpurun -- bash tools/probes/conv_bounds.sh run
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/../.. && pwd)}
PKG=$ROOT/eccv2022-multi-scale-and-cross-scale-contrastive-segmentation_amd
OUT=$ROOT/tools/probes/variants
VARIANTS="${VARIANTS:-1 2 4 8 3 6 7 15}"
if [ "$1" = wbuild ]; then
  mkdir -p $OUT
  for v in ${WVARIANTS:-1 2 3}; do
    ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DDCL_WG_PROBE=$v -c $PKG/csrc/dcl_wgrad3x3d.hip -o $OUT/wg_$v.o &&
      hipcc -shared -fPIC --offload-arch=gfx950 $(ls $PKG/csrc/build/*.o | grep -v dcl_wgrad3x3d.o | grep -v wgrad3x3s | grep -v tokgemm) $OUT/wg_$v.o -o $OUT/libdcl_wprobe_$v.so ) &
  done
  wait
  ls -la $OUT/*wprobe*.so
elif [ "$1" = wrun ]; then
  cd $ROOT
  echo "variant 0 (product)"; python3 tools/probes/wgrad_bounds.py
  for v in ${WVARIANTS:-1 2 3}; do
    echo "variant $v"; DCL_LIB_PATH=$OUT/libdcl_wprobe_$v.so python3 tools/probes/wgrad_bounds.py
  done
elif [ "$1" = build ]; then
  mkdir -p $OUT
  for v in $VARIANTS; do
    ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DDCL_CONV_PROBE=$v -c $PKG/csrc/dcl_conv3x3.hip -o $OUT/conv_$v.o &&
      hipcc -shared -fPIC --offload-arch=gfx950 $(ls $PKG/csrc/build/*.o | grep -v dcl_conv3x3.o | grep -v wgrad3x3s | grep -v tokgemm) $OUT/conv_$v.o -o $OUT/libdcl_probe_$v.so ) &
  done
  wait
  ls -la $OUT/*.so
else
  cd $ROOT
  echo "variant 0 (product)"; python3 tools/probes/conv_bounds.py
  for v in $VARIANTS; do
    echo "variant $v"; DCL_LIB_PATH=$OUT/libdcl_probe_$v.so python3 tools/probes/conv_bounds.py
  done
fi
