#!/bin/bash
# Cost of building without packed FP32: alternating runs of the product library and tools/probes/variants/libdcl_packed.so (the
# same sources built WITH packed FP32: make -C <pkg>/csrc NOPK= OUT=../../tools/probes/variants/libdcl_packed.so OBJDIR=/tmp/pk_obj)
# on one box:   gpurun -- bash tools/probes/ab_packed.sh
cd ${GRAFT_REPO_ROOT:-.}
P=$PWD/tools/probes/variants/libdcl_packed.so
one() { python bench.py --steps $1 --warmup 3 --no-cpu-baseline --no-eager-step "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('contrastive_loss_fwd_bwd_ms'))"; }
for i in 1 2; do
  for what in "--workload loss" "--config 4" "--config 5"; do
    echo "unpacked $what: $(one 10 $what)"
    echo "packed   $what: $(DCL_LIB_PATH=$P one 10 $what)"
  done
done
