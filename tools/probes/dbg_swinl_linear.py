"""Is the whole-model gradient of the Swin-L FPN train fixture stable under fp32-level perturbations of the stage-1
Linear outputs?  (Round 3: moving those Linears to dcl_gemm_f16x3 -- closer to float64 in every product -- moved the
model's dx error from 5e-3 to 2e-2 of max.)  Runs the fixture with the library Linears, then with their outputs
multiplied by (1 + eps * gaussian) for eps = 1e-7 (below one fp32 ulp on average) and 3e-7."""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import __graft_entry__  # noqa: F401,E402
import test_models as tm  # noqa: E402
from mscs_amd.models import ops  # noqa: E402

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "G11_train_upernet_swinL_fpn"


def run(label):
    e = tm._train_errors(name, dev, "f64_")
    print(label, {k: round(v, 6) for k, v in e.items() if k in ("loss", "dx", "pgrad", "pgrad_first4", "running")}, flush=True)


run("f16x3 Linears")
ops.TokenLinear.f16x3 = False
run("library Linears")
orig = ops.TokenLinear.forward
for eps in (1e-7, 3e-7):
    for seed in (0, 1, 2):
        def fwd(self, x, eps=eps):
            y = torch.nn.Linear.forward(self, x)
            if x.numel() // x.shape[-1] >= 1024:
                y = y * (1 + eps * torch.randn_like(y))
            return y
        torch.manual_seed(seed)
        ops.TokenLinear.forward = fwd
        run(f"library Linears, outputs perturbed by {eps:g} (seed {seed})")
ops.TokenLinear.forward = orig
