#!/usr/bin/env python3
"""Dump the reference's dataset class tables (utils/datasets_info/*.py: CLASS_INFO =
[remap, names, categories] per experiment) as a JSON DATA file for mscs_amd.utils.datasets_info.
Build-container only (imports /root/reference through tools/ref_shim.py)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shim  # noqa: E402

ref_shim.install()
ref_shim.quiet()
from utils import DATASETS_INFO  # noqa: E402


def jsonable(x):
    if isinstance(x, dict):
        return {"__dict__": [[jsonable(k), jsonable(v)] for k, v in x.items()]}
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    if isinstance(x, (int, float, str, bool)) or x is None:
        return x
    return str(x)


out = {}
for ds in DATASETS_INFO:
    out[ds] = [[jsonable(part) for part in ci] for ci in DATASETS_INFO[ds].CLASS_INFO]
dst = os.path.join(os.path.dirname(__file__), "..",
                   "eccv2022-multi-scale-and-cross-scale-contrastive-segmentation_amd", "utils",
                   "datasets_info.json")
with open(dst, "w") as f:
    json.dump(out, f, separators=(",", ":"))
print("wrote", dst, os.path.getsize(dst), "bytes")
