"""Whole-model train fixture of UPerNet + Swin-T with and without the fusion convolution's split: every error metric."""
import json
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import __graft_entry__  # noqa: F401,E402
import test_models as tm  # noqa: E402
from mscs_amd.models import UPerNet as U  # noqa: E402,F401
import importlib  # noqa: E402

um = importlib.import_module("mscs_amd.models.UPerNet")
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "G11_train_upernet_swinT_fpn"
z = np.load(os.path.join(tm.GOLDEN, name + ".npz"))
print("reference loss f64", float(z["f64_loss"]), "f32", float(z["loss"]))


def show(label, e):
    print(label, {k: ([round(v, 7) for v in e[k]] if isinstance(e[k], list) else round(e[k], 7))
                  for k in ("out", "loss", "dx", "pgrad", "running")}, flush=True)


show("hip, split", tm._train_errors(name, dev, "f64_"))
orig = um.FPN.__init__


def init(self, config, experiment):
    orig(self, config, experiment)
    self.head_split = False


um.FPN.__init__ = init
show("hip, materialised", tm._train_errors(name, dev, "f64_"))
show("stock kernels", tm._train_errors(name, dev, "f64_", library=True))
