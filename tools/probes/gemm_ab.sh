#!/bin/bash
# A/B builds of csrc/dcl_gemm.hip on one box:   (here)     bash tools/probes/gemm_ab.sh build "NAME=-DFLAG=1 ..." ...
#                                               (GPU box)  gpurun -- bash tools/probes/gemm_ab.sh run [shapes]
# Each variant is a small shared library (dcl_gemm.hip + dcl_capi.cpp) under tools/probes/_build/gemm_<NAME>.so;
# gemm_ab.py loads all of them into ONE process and times them alternately (medians), so clock drift hits all alike.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/../.. && pwd)}
PKG=$ROOT/eccv2022-multi-scale-and-cross-scale-contrastive-segmentation_amd
OUT=$ROOT/tools/probes/_build
if [ "$1" = build ]; then
  shift
  mkdir -p $OUT
  for spec in "$@"; do
    name=${spec%%=*}; flags=${spec#*=}; [ "$flags" = "$spec" ] && flags=""
    ( hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -shared $PKG/csrc/dcl_gemm.hip -x hip $PKG/csrc/dcl_capi.cpp -o $OUT/gemm_$name.so && echo built $name ) &
  done
  wait
else
  shift || true
  cd $ROOT
  python3 tools/probes/gemm_ab.py "$@"
fi
