"""G13 UPerNet-FPN fixture twice in one process: tap products on the split-f16 GEMM vs on the library; compare what enters and leaves
the head split's backward."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_models as tm
from conftest import GOLDEN
from mscs_amd.models import ops
name = "G13_module_upernet_fpn"
dev = torch.device("cuda:0")
z = np.load(os.path.join(GOLDEN, name + ".npz"))
rec = {}
orig_bwd = ops._HeadSplit.backward
def run(gemm):
    ops._CoarseTaps.gemm = gemm
    torch.manual_seed(0)
    mod = tm._module_under_test(name, dev)
    tm.fill_state_dict_(mod)
    mod.train().to(dev)
    xs = [tm.model_input(tuple(int(v) for v in sh), seed=11 + i).to(dev).requires_grad_(True) for i, sh in enumerate(z["input_shapes"])]
    cap = {}
    def bwd(ctx, gy):
        cap["gy"] = gy.detach().clone()
        r = orig_bwd(ctx, gy)
        cap["gw"] = r[5].detach().clone()
        cap["rest"] = [t.detach().clone() if isinstance(t, torch.Tensor) else None for t in r]
        return r
    ops._HeadSplit.backward = staticmethod(bwd)
    orig_conv = ops.conv3x3_over_upsampled
    def conv(ts, *a_, **k_):
        y = orig_conv(ts, *a_, **k_)
        cap["y"] = y.detach().clone()
        cap["ts"] = [t.detach().clone() for t in ts]
        return y
    import mscs_amd.models.UPerNet as um
    ops.conv3x3_over_upsampled = conv
    outs = tm._flatten(mod(list(xs)))
    tm._probe_loss(outs).backward()
    cap["outs"] = [o.detach().clone() for o in outs]
    ops.conv3x3_over_upsampled = orig_conv
    cap["wgrad"] = dict((k, p.grad.clone()) for k, p in mod.named_parameters())
    return cap
a, b = run(True), run(False)
rel = lambda x, y: ((x - y).abs().max() / y.abs().max()).item()
print("outs", [rel(x, y) for x, y in zip(a["outs"], b["outs"])])
print("y of head split", rel(a["y"], b["y"]), "inputs", [rel(x, y) for x, y in zip(a["ts"], b["ts"])])
print("gy into head split", rel(a["gy"], b["gy"]), "max", a["gy"].abs().max().item(), b["gy"].abs().max().item(),
      "mean abs", a["gy"].abs().mean().item(), b["gy"].abs().mean().item())
print("gw from head split", rel(a["gw"], b["gw"]))
for lo, nm in ((0, "P2"), (256, "P5"), (512, "P4"), (768, "P3")):
    print("   slice", nm, rel(a["gw"][:, lo:lo + 256], b["gw"][:, lo:lo + 256]))
for k in a["wgrad"]:
    r = rel(a["wgrad"][k], b["wgrad"][k])
    if r > 1e-4:
        print("param", k, r)
ga, gb = a["gy"], b["gy"]
print("gy shapes / strides", ga.shape, ga.stride(), gb.shape, gb.stride())
d = (ga - gb).abs()
idx = d.flatten().argmax().item()
n_, c_, h_, w_ = np.unravel_index(idx, ga.shape)
print("largest difference at", (n_, c_, h_, w_), ga[n_, c_, h_, w_].item(), gb[n_, c_, h_, w_].item())
print("per-channel max |diff| / max:", (d.amax((0, 2, 3)) / gb.abs().max()).topk(8))
print("fraction of elements differing by > 1e-3 of max:", (d > 1e-3 * gb.abs().max()).float().mean().item())
print("sorted equal:", torch.allclose(ga.flatten().sort().values, gb.flatten().sort().values, rtol=1e-4, atol=1e-12))
