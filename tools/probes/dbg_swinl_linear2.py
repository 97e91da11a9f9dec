"""Per-call accuracy of the split-f16 Linear products on the REAL tensors of the Swin-L FPN train fixture (vs float64 of
the same inputs), next to the library's fp32 GEMM."""
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
f_fwd, f_dg, f_wg = ops.linear_f16x3, ops.linear_dgrad_f16x3, ops.linear_wgrad_f16x3


def rel(a, r):
    return ((a.double() - r).abs().max() / r.abs().max().clamp_min(1e-300)).item()


def stats(t):
    a = t.abs()
    return f"max {a.max().item():.3e} mean {a.mean().item():.3e} zeros {(t == 0).float().mean().item():.3f}"


def fwd(x2, w, b=None, tag_out=True):
    y = f_fwd(x2, w, b, tag_out)
    r = x2.double() @ w.double().t() + (b.double() if b is not None else 0)
    print(f"fwd   {tuple(x2.shape)}x{tuple(w.shape)}: f16x3 {rel(y, r):.2e} lib {rel(torch.nn.functional.linear(x2, w, b), r):.2e} | x {stats(x2)}")
    return y


def dg(gy2, w):
    g = f_dg(gy2, w)
    r = gy2.double() @ w.double()
    print(f"dgrad {tuple(gy2.shape)}x{tuple(w.shape)}: f16x3 {rel(g, r):.2e} lib {rel(gy2.mm(w), r):.2e} | gy {stats(gy2)}")
    return g


def wg(gy2, x2):
    g = f_wg(gy2, x2)
    r = gy2.double().t() @ x2.double()
    print(f"wgrad {tuple(gy2.shape)}^T x{tuple(x2.shape)}: f16x3 {rel(g, r):.2e} lib {rel(gy2.t().mm(x2), r):.2e}")
    return g


ops.linear_f16x3, ops.linear_dgrad_f16x3, ops.linear_wgrad_f16x3 = fwd, dg, wg
e = tm._train_errors(name, dev, "f64_")
print({k: round(v, 6) for k, v in e.items() if k in ("loss", "dx", "pgrad", "pgrad_first4", "running")})
