"""Time of the stem's weight gradient (3 -> 64, stride 2, 12 x 512 x 1024): padded split-f16 path against the library's kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd  # noqa: E402,F401
from mscs_amd.models import ops  # noqa: E402

dev = torch.device("cuda:0")
x = torch.randn(12, 3, 512, 1024, device=dev)
gy = torch.randn(12, 64, 256, 512, device=dev)
w = torch.randn(64, 3, 3, 3, device=dev)


def padded():
    xp = x.new_zeros((12, 16, 512, 1024))
    xp[:, :3] = x
    return ops.conv3x3_wgrad(xp, gy, 2)[:, :3].contiguous()


def library():
    return torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]


fns = {"padded": padded, "library": library}
if hasattr(ops, "stem_wgrad"):
    fns["dedicated"] = lambda: ops.stem_wgrad(x, gy)
ref = library().double()
for name, fn in fns.items():
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    print(f"{name:10s} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us   (distance to the library's result {err:.1e})")
