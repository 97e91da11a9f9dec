import json
for n in ("default", "blocky", "config4", "config5"):
    d = json.loads(open(f"gpurun_out/r03_bench_{n}.json").read().strip().splitlines()[-1])
    print(n, d["ms_per_step"], d["value"], d.get("eager_gpu_step_ms"), d.get("speedup_vs_eager_gpu_step"),
          d.get("eager_gpu_step_ms_miopen_find"), d["roofline"]["frac"], d["roofline"].get("launch_ms"),
          (d.get("cpu_baseline") or {}).get("value"))
