#!/bin/bash
# Alternating A/B of bench.py on one box:  gpurun -- bash tools/probes/bench_ab.sh "ENV_A" "ENV_B" [rounds] [bench args...]
#   e.g. bash tools/probes/bench_ab.sh "DCL_GEMM_CONV1X1=1" "DCL_GEMM_CONV1X1=0" 3
A="$1"; B="$2"; R=${3:-3}; shift 3 || true
cd ${GRAFT_REPO_ROOT:-.}
for i in $(seq 1 $R); do
  for v in "$A" "$B"; do
    ms=$(env $v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-eager-step "$@" 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$v $ms"
  done
done
