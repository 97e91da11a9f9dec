"""Window attention forward / backward per launch, vector-ALU kernels (mask 0) against the matrix-core kernels (mask 3),
on the Swin-L (640^2, batch 16) and Swin-T (512^2, batch 16) stage shapes; max |difference| of the outputs."""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import __graft_entry__  # noqa: F401,E402
from mscs_amd import _lib  # noqa: E402
from mscs_amd.models import ops  # noqa: E402

L = _lib.lib()
dev = "cuda"


def t(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [("swinL s1", 16, 160, 160, 192, 6), ("swinL s2", 16, 80, 80, 384, 12), ("swinL s3", 16, 40, 40, 768, 24),
          ("swinL s4", 16, 20, 20, 1536, 48), ("swinT s1", 16, 128, 128, 96, 3), ("swinT s3", 16, 32, 32, 384, 12)]
only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, B, H, W, C, heads in shapes:
    if only and only not in name:
        continue
    for shift in (0, 3):
        torch.manual_seed(0)
        qkv = torch.randn(B, H * W, 3 * C, device=dev, requires_grad=True)
        qb = torch.randn(3 * C, device=dev) * 0.1
        bias = torch.randn(heads, 49, 49, device=dev) * 0.5
        gy = torch.randn(B, H * W, C, device=dev)
        res = {}
        for mask in (0, 3):
            L.dcl_winattn_set_mfma(mask)
            out = ops.window_attention(qkv, qb, bias, H, W, heads, shift, 32 ** -0.5)
            tf = t(lambda: ops.window_attention(qkv.detach(), qb, bias, H, W, heads, shift, 32 ** -0.5))
            qkv.grad = None
            out.backward(gy)
            g = qkv.grad.clone()

            def fb():
                o = ops.window_attention(qkv, qb, bias, H, W, heads, shift, 32 ** -0.5)
                o.backward(gy)
            tfb = t(fb)
            res[mask] = (out.detach(), g, tf, tfb - tf)
        L.dcl_winattn_set_mfma(3)
        do = (res[0][0] - res[3][0]).abs().max().item() / res[0][0].abs().max().item()
        dg = (res[0][1] - res[3][1]).abs().max().item() / res[0][1].abs().max().item()
        print(f"{name} shift {shift}: fwd {res[0][2]:7.1f} -> {res[3][2]:7.1f} us   bwd {res[0][3]:7.1f} -> {res[3][3]:7.1f} us   "
              f"rel diff out {do:.1e} grad {dg:.1e}", flush=True)
