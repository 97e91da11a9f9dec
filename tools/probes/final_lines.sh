python -m pytest tests -q -m gpu 2>&1 | tail -1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > gpurun_out/r03_bench_default.json 2>/dev/null
python bench.py --labels blocky --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_bench_blocky.json
python bench.py --config 4 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_bench_config4.json
python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_bench_config5.json
python tools/probes/print_bench.py
