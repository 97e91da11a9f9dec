"""Run-to-run reproducibility of an HRNet exchange module under the merged-branch schedule (models/merged.py).

    DBG_HW=128,256 python tools/probes/dbg_merged.py [whole] [on|off|alt] [interleave] [runs=N]

Runs the module (or only its branches) 8 times on the same inputs and reports which outputs / gradients differ from the first run
(`on`: merged schedule, `off`: one stream per branch, `alt`: alternating).  Found with it: issuing branch 0's blocks BETWEEN the
coarse group's depths (alternating streams in the backward) made single norm-backward outputs differ from run to run; the shipped
order (branch 0 after the coarse chain) is clean (HRNet.py _run_branches_merged)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mscs_amd  # noqa: F401,E402
from test_merged_branches import _module  # noqa: E402
from mscs_amd.models import fused_bn  # noqa: E402

dev = torch.device("cuda:0")
hm, mod, ch = _module(4, dev)
state = {k: v.clone() for k, v in mod.state_dict().items()}
hw = tuple(int(v) for v in os.environ.get("DBG_HW", "64,96").split(","))
xs0 = [torch.randn(2, c, hw[0] >> i, hw[1] >> i, device=dev) for i, c in enumerate(ch)]
whole = "whole" in sys.argv
hm._MERGE_INTERLEAVE = "interleave" in sys.argv
nrun = next((int(a[5:]) for a in sys.argv if a.startswith("runs=")), 8)
seq = [False, True] * nrun if "alt" in sys.argv else ([False] * nrun if "off" in sys.argv else [True] * nrun)
res = []
for merged in seq:
    mod.load_state_dict(state)
    mod.zero_grad(set_to_none=True)
    xs = [x.clone().requires_grad_(True) for x in xs0]
    if "group" in sys.argv:         # the stacked-exchange schedule (fused_bn.FORCE_GROUP) against the free-running one
        fused_bn.FORCE_GROUP = merged
    else:
        hm._MERGE_BRANCHES = merged
    try:
        outs = mod(list(xs)) if whole else mod._run_branches(list(xs))
    finally:
        fused_bn.FORCE_GROUP = False
    sum((o * torch.cos(torch.arange(o.numel(), device=dev).view(o.shape) * 0.37)).mean() for o in outs).backward()
    torch.cuda.synchronize()
    names = [f"out{i}" for i in range(len(outs))] + [f"xgrad{i}" for i in range(4)] + [n for n, p in mod.named_parameters() if p.grad is not None]
    vals = [o.detach().clone() for o in outs] + [x.grad.clone() for x in xs] + [p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None]
    res.append(dict(zip(names, vals)))
for j in range(1, len(res)):
    bad = [(n, ((res[0][n] - res[j][n]).abs().max() / (res[0][n].abs().max() + 1e-30)).item()) for n in res[0] if not torch.equal(res[0][n], res[j][n])]
    nbad = nbad + (1 if bad else 0) if j > 1 else (1 if bad else 0)
    if bad or "-v" in sys.argv:
        print(f"run {j} (merged={seq[j]}): {len(bad)} of {len(res[0])} tensors differ", [(n.replace('branches.', ''), f"{e:.0e}") for n, e in bad[:8]])
print(f"SUMMARY lib={os.environ.get('DCL_LIB_PATH', 'product')} args={sys.argv[1:]}: {nbad} of {len(res) - 1} runs differ from run 0")
