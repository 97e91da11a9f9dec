#!/bin/bash
# The round's bench lines + suite + smoke on one box (run through gpurun):  gpurun --timeout 2400 -- bash tools/probes/final_lines.sh r05
T=${1:-r05}
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests -q -m gpu 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --detail-file gpurun_out/${T}_bench_default_detail.json 2>/dev/null | tail -1 > gpurun_out/${T}_bench_default.json
python bench.py --workload loss 2>/dev/null | tail -1 > gpurun_out/${T}_bench_loss.json
python bench.py --labels blocky --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_blocky.json
python bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_config4.json
python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${T}_bench_config5.json
for f in default loss blocky config4 config5; do python - <<PY
import json
d = json.load(open("gpurun_out/${T}_bench_$f.json"))
print("$f", d["ms_per_step"], d["value"], d.get("contrastive_loss_fwd_bwd_ms"), (d.get("roofline") or {}).get("frac"), len(json.dumps(d)))
PY
done
