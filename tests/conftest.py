import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["config"] = json.loads(str(d["config_json"]))
    return d


def golden_names(prefixes):
    out = []
    for f in sorted(os.listdir(GOLDEN)):
        if f.endswith(".npz") and any(f.startswith(p) for p in prefixes):
            out.append(f[:-4])
    return out


def num_classes_for(cfg):
    """K = num_all_classes for the fixture's dataset/experiment."""
    if "_override_num_all_classes" in cfg:
        return int(cfg["_override_num_all_classes"])
    assert cfg["dataset"] == "CITYSCAPES" and cfg["experiment"] == 1
    return 20


@pytest.fixture(scope="session")
def oracle():
    from oracle import dcl_oracle
    dcl_oracle.build_c_oracle()
    return dcl_oracle
