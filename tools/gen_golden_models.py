#!/usr/bin/env python3
"""G7: model parity fixtures from the REFERENCE models (build container only).

For each model: the state_dict manifest (key -> shape), and forward outputs in eval mode at a tiny
input with name-seeded weights (tools/model_fill.py): float64 checksums plus a strided sample."""
import builtins
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shim  # noqa: E402
from model_fill import fill_state_dict_, model_input  # noqa: E402

ref_shim.install()
ref_shim.quiet()
_print = builtins.print
builtins.print = lambda *a, **k: None
from models import HRNet, UPerNet  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")

CASES = {
    "G7_hrnet48_ms4": dict(cls="HRNet", shape=(1, 3, 64, 128), cfg={
        'backbone': 'hrnet48', 'pretrained': False, 'dataset': 'CITYSCAPES', 'align_corners': True, 'out_stride': 4,
        'ms_projector': {'mlp': [[1, -1, 1]], 'scales': 4, 'd': 256, 'use_bn': True, 'before_context': True}}, exp=1),
    "G7_hrnet48_ms3": dict(cls="HRNet", shape=(1, 3, 64, 64), cfg={
        'backbone': 'hrnet48', 'pretrained': False, 'dataset': 'CITYSCAPES', 'align_corners': True, 'out_stride': 4,
        'ms_projector': {'mlp': [[1, -1, 1]], 'scales': 3, 'd': 256, 'use_bn': True}}, exp=1),
    "G7_hrnet48_single": dict(cls="HRNet", shape=(1, 3, 64, 64), cfg={
        'backbone': 'hrnet48', 'pretrained': False, 'dataset': 'CITYSCAPES', 'align_corners': False, 'out_stride': 4,
        'projector': {'mlp': [[1, -1, 1]], 'd': 128, 'use_bn': True}}, exp=1),
    "G7_upernet_swinT_fpn": dict(cls="UPerNet", shape=(1, 3, 64, 64), cfg={
        'backbone': 'swinT', 'pretrained': False, 'dataset': 'ADE20K', 'align_corners': False, 'out_stride': 4,
        'aux_head': {'in_index': 2, 'dropout_rate': 0.1, 'out_channels': 256}, 'dropout_rate': 0.1,
        'ms_projector': {'mlp': [[1, -1, 1]], 'scales': 4, 'd': 256, 'use_bn': True, 'position': 'fpn'}}, exp=1),
    "G7_upernet_swinT_backbone": dict(cls="UPerNet", shape=(1, 3, 64, 96), cfg={
        'backbone': 'swinT', 'pretrained': False, 'dataset': 'ADE20K', 'align_corners': False, 'out_stride': 4,
        'ms_projector': {'mlp': [[1, -1, 1]], 'scales': 4, 'd': 256, 'use_bn': True, 'position': 'backbone'}}, exp=1),
    "G7_upernet_swinL_fpn": dict(cls="UPerNet", shape=(1, 3, 64, 64), cfg={
        'backbone': 'swinL', 'pretrained': False, 'dataset': 'ADE20K', 'align_corners': False, 'out_stride': 4,
        'aux_head': {'in_index': 2, 'dropout_rate': 0.1, 'out_channels': 256},
        'ms_projector': {'mlp': [[1, -1, 1]], 'scales': 4, 'd': 256, 'use_bn': True, 'position': 'fpn'}}, exp=1),
}


def flatten(out):
    res = []
    if isinstance(out, (list, tuple)):
        for o in out:
            res += flatten(o)
    elif torch.is_tensor(out):
        res.append(out)
    return res


def main():
    only = sys.argv[1:]
    for name, case in CASES.items():
        if only and not any(o in name for o in only):
            continue
        cfg = json.loads(json.dumps(case["cfg"]))
        try:
            model = {"HRNet": HRNet, "UPerNet": UPerNet}[case["cls"]](config=cfg, experiment=case["exp"])
        except Exception as e:  # noqa: BLE001
            _print(f"{name}: reference construction failed: {type(e).__name__}: {e}")
            continue
        fill_state_dict_(model)
        model.eval()
        x = model_input(case["shape"])
        with torch.no_grad():
            outs = flatten(model(x))
        d = {"config_json": np.array(json.dumps(case["cfg"])), "experiment": np.int32(case["exp"]),
             "input_shape": np.array(case["shape"], dtype=np.int32), "n_outputs": np.int32(len(outs)),
             "manifest_json": np.array(json.dumps({k: list(v.shape) for k, v in model.state_dict().items()})),
             "torch_version": np.array(torch.__version__)}
        for i, o in enumerate(outs):
            d[f"out{i}_shape"] = np.array(o.shape, dtype=np.int32)
            d[f"out{i}_sum"] = np.float64(o.double().sum().item())
            d[f"out{i}_abs_sum"] = np.float64(o.double().abs().sum().item())
            flat = o.flatten()
            step = max(1, flat.numel() // 4096)
            d[f"out{i}_sample"] = flat[::step].numpy().copy()
            d[f"out{i}_step"] = np.int64(step)
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **d)
        _print(f"wrote {path} ({os.path.getsize(path) // 1024} KiB), outputs: {[tuple(o.shape) for o in outs]}")


if __name__ == "__main__":
    main()
