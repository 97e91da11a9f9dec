set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -x > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
bash tools/collect_profiles.sh r03 > gpurun_out/collect.log 2>&1
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/bench_default.err
python bench.py --labels blocky --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_bench_blocky.json
python bench.py --workload loss --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_bench_loss.json
python bench.py --config 4 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_bench_config4.json
python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_bench_config5.json
python tools/host_vs_gpu.py --lead > gpurun_out/r03_host_vs_gpu.txt 2>&1
python - <<'P'
import json
for n in ("default","blocky","loss","config4","config5"):
    try:
        d=json.loads(open(f"gpurun_out/r03_bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["ms_per_step"], d["value"], d.get("eager_gpu_step_ms"), d.get("speedup_vs_eager_gpu_step"), d.get("eager_gpu_step_ms_miopen_find"), d["roofline"]["frac"], d["roofline"].get("launch_ms"))
    except Exception as e: print(n, "ERR", e)
P
cat gpurun_out/r03_host_vs_gpu.txt | tail -3
