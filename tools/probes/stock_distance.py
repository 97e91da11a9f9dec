"""Distance of the STOCK ATen / MIOpen kernels (fp32, this GPU) and of the HIP path to the reference's fp64 record, per G13 fixture
and metric of tests/test_models.py::_module_errors -- the yardstick for that test's fixed bars.

    python tools/probes/stock_distance.py [fixture ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_models as T  # noqa: E402

orig = T._module_under_test
torch.backends.cudnn.allow_tf32 = False
torch.backends.cuda.matmul.allow_tf32 = False
names = sys.argv[1:] or ["G13_module_layer1", "G13_module_upernet_fpn", "G13_module_stage3", "G13_module_stage4", "G13_module_fuse_chain"]
for name in names:
    for label, build in (("stock", lambda n, d: orig(n, torch.device("cpu"))), ("hip  ", orig)):
        T._module_under_test = build          # "stock": the block as built for the CPU (plain nn modules), run on the GPU
        try:
            r = T._module_errors(name, torch.device("cuda:0"), "f64_")
        finally:
            T._module_under_test = orig
        print(label, name, {k: (f"{max(v):.2e}" if isinstance(v, list) else (f"{v:.2e}" if isinstance(v, float) else v)) for k, v in r.items()})
