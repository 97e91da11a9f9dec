"""bench.py's stdout contract: ONE compact JSON line (<= 4 KB) that the driver's ~9 KB stdout tail can hold (a 23.7 KB
line was recorded as `parsed: null`), carrying BASELINE.json's metric, one `roofline`, `roofline_step` and a `cpu_baseline`; the
per-shape tables and notes go to the detail record."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _canned_rows(n_mfma=14, n_hbm=16):
    rows = []
    for i in range(n_mfma):
        shapes = {f"12x{48 << (j % 4)}->{48 << (j % 4)}x{128 >> (j % 4)}x{256 >> (j % 4)}": [40 + j, 3.1 + j, 5.0e12 + j, 8.0e8] for j in range(9)}
        rows.append({"bound": "mfma", "kernel": f"k_conv3x3_il<3,{i}>", "entry": "dcl_conv3x3_f16x3", "calls": 242 - i, "total_ms": 21.3 - i,
                     "flops": 5.381e12, "bytes": 1.9e10, "achieved": 252.7, "peak": 833.3, "unit": "TFLOP/s", "frac": 0.3032, "shapes": shapes})
    for i in range(n_hbm):
        shapes = {f"12x{48 << (j % 4)}x{32768 >> (2 * (j % 4))}": [8, 0.3 + j, 0.0, 9.0e8] for j in range(8)}
        rows.append({"bound": "hbm", "kernel": f"k_bn_bwd_apply<{i}>", "entry": "dcl_bn_bwd_apply", "calls": 150, "total_ms": 10.8,
                     "flops": 0.0, "bytes": 5.2e10, "achieved": 4961.2, "peak": 8000.0, "unit": "GB/s", "frac": 0.62, "shapes": shapes})
    return rows


class _Args:
    batch, height, width, scales, config = 12, 512, 1024, 3, 2
    no_cross, labels, classes = False, "iid", 20
    kernel_table = None
    detail_file = None


def _full_record(bench):
    args = _Args()
    out = {"metric": "train_images_per_sec", "value": 136.612, "unit": "images/s", "n_gpus": 1, "steps": 20, "warmup": 5,
           "ms_per_step": 87.84, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (f16x3-emulated)",
           "data": "synthetic", "config": {"workload": bench.workload_name(args, "step"), "terms": [[0, 0, 9804, 9804]] * 5},
           "contrastive_loss_fwd_bwd_ms": 4.56, "metrics_in_step": True, "lazy_logits": True, "lazy_projector": True,
           "fused_optimizer": True, "config_keys_beyond_reference": [], "peak_mem_gb": 30.9}
    out.update(bench.roofline_from_rows(_canned_rows(), args))
    out["roofline_other"] += [{"bound": "mfma", "kernel": "k_sweep<MODE_BWD, stream-K> (dcl_infonce_bwd_streamk, 256 persistent workgroups), "
                               "both products in f16x3", "achieved": 311.0, "peak": 833.3, "unit": "TFLOP/s", "frac": 0.3732,
                               "peak_note": "x" * 300, "traffic": 197.0e6, "traffic_source": "y" * 400, "algorithmic_bytes": 30117888,
                               "launch_ms": 0.3163}]
    fl = out.pop("_step_flops")
    out["roofline_step"] = {"bound": "mfma", "kernel": "all matrix-pipe launches of one training step " + "z" * 200, "algorithmic_flops": fl,
                            "achieved": 176.0, "peak": 833.3, "unit": "TFLOP/s", "frac": 0.2112, "note": "n" * 300}
    out["cpu_baseline"] = {"value": 0.16086, "unit": "images/s", "cores": 64, "kind": "port", "sample": "s" * 700,
                           "sample_short": "one whole step on the host", "model_seconds": 37.1, "loss_seconds": 37.5, "sample_seconds": 74.6}
    out.update({"plain_config_ms_per_step": 90.5, "plain_config_keys": ["fused_optimizer", "lazy_logits", "lazy_projector"],
                "eager_gpu_step_ms": 617.7, "speedup_vs_eager_gpu_step": 7.03, "eager_gpu_step_note": "e" * 420,
                "eager_gpu_step_ms_miopen_find": 618.0, "speedup_vs_eager_gpu_step_miopen_find": 7.04})
    return out, args


def test_compact_line_fits_the_driver_tail_and_carries_the_contract_keys():
    bench = _bench()
    full, args = _full_record(bench)
    assert len(json.dumps(full)) > 3 * bench.LINE_LIMIT          # the canned record is as fat as round 4's
    line = bench.compact_line(full)
    assert len(line) < 4096 and "\n" not in line
    got = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "contrastive_loss_fwd_bwd_ms", "roofline", "roofline_step", "cpu_baseline", "eager_gpu_step_ms",
              "speedup_vs_eager_gpu_step", "plain_config_ms_per_step"):
        assert k in got, k
    assert set(got["config"]) == {"workload"}
    r = got["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launches", "step_ms", "algorithmic_flops", "traffic"):
        assert k in r, k
    assert "shapes" not in r and "note" not in r and "traffic_source" not in r
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    cb = got["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample", "sample_seconds"} <= set(cb) and len(cb["sample"]) <= 200
    assert got["roofline_sweep"]["kernel"].startswith("k_sweep")
    assert "roofline_other" not in got and "roofline_hbm" not in got


def test_emit_prints_the_compact_line_last_on_stdout_and_detail_on_stderr(capsys, tmp_path):
    bench = _bench()
    full, args = _full_record(bench)
    args.detail_file = str(tmp_path / "detail.json")
    bench.emit(full, args)
    cap = capsys.readouterr()
    lines = cap.out.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096
    assert json.loads(lines[0])["detail"] == args.detail_file
    assert cap.err.startswith("bench-detail: ")
    detail = json.loads(open(args.detail_file).read())
    assert "roofline_hbm" in detail and "shapes" in detail["roofline"]


def test_compact_line_sheds_optional_keys_rather_than_overflowing():
    bench = _bench()
    full, args = _full_record(bench)
    full["config"]["workload"] = "w" * 5000          # (is cut to 300 characters)
    full["roofline"]["kernel"] = "k" * 5000
    line = bench.compact_line(full)
    assert len(line) < 4096 and json.loads(line)["value"] == 136.612
