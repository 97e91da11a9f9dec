#!/usr/bin/env python3
"""G12: per-step metric fixtures from the REFERENCE (build container only): t_get_confusion_matrix,
t_get_pixel_accuracy, t_get_mean_iou (utils/torch_utils.py:157-283) on small logits / targets, with and without
an ignore class, with ties and an already-accumulated matrix."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shim  # noqa: E402

ref_shim.install()
ref_shim.quiet()
from utils import t_get_confusion_matrix, t_get_mean_iou, t_get_pixel_accuracy  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")


def case(dataset, experiment, C, K, n, H, W, seed, ties):
    gen = torch.Generator().manual_seed(seed)
    logits = torch.randn(n, C, H, W, generator=gen)
    if ties:        # exact ties between classes: argmax must pick the first
        logits = torch.round(logits * 2) / 2
    target = torch.randint(0, K, (n, H, W), generator=gen)
    cm = t_get_confusion_matrix(logits, target, dataset)
    cm2 = t_get_confusion_matrix(logits.flip(0), target, dataset, existing_matrix=cm.clone())
    pa, pac = t_get_pixel_accuracy(cm)
    miou = t_get_mean_iou(cm, experiment, dataset)["mean_iou"]
    return {"logits": logits.numpy(), "target": target.numpy().astype(np.int32), "cm": cm.numpy(),
            "cm_accumulated": cm2.numpy(), "pa": np.float32(pa.item()), "pac": np.float32(pac.item()),
            "miou": np.float32(miou.item()), "dataset": np.array(dataset), "experiment": np.int32(experiment)}


def main():
    d = {}
    for name, c in {
        "cts": case("CITYSCAPES", 1, 19, 20, 2, 24, 40, 3, False),           # ignore id 19 in the target
        "cts_ties": case("CITYSCAPES", 1, 19, 20, 1, 16, 24, 4, True),
        "ade": case("ADE20K", 1, 150, 151, 1, 32, 32, 5, False),
        "cadis_noignore": case("CADIS", 1, 8, 8, 2, 16, 16, 6, False),       # experiment without an ignore class
    }.items():
        for k, v in c.items():
            d[f"{name}__{k}"] = v
    d["torch_version"] = np.array(torch.__version__)
    path = os.path.join(OUT, "G12_metrics.npz")
    np.savez_compressed(path, **d)
    print(f"wrote {path} ({os.path.getsize(path) // 1024} KiB)")


if __name__ == "__main__":
    main()
