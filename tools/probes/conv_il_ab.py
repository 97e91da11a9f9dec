"""A/B of the interleaved-staging convolution tiles against the fenced ones on the step's shapes (forward and data gradient
are the same kernel) + correctness against fp64."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd  # noqa: E402,F401
from mscs_amd import _lib  # noqa: E402
from mscs_amd.models import ops  # noqa: E402
from mscs_amd.models.amax import amax_of  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [(12, 48, 48, 128, 256), (12, 96, 96, 64, 128), (12, 192, 192, 32, 64), (12, 384, 384, 16, 32), (12, 64, 64, 128, 256),
          (12, 720, 720, 128, 256), (2, 512, 512, 128, 128)]
for (n, ci, co, h, w) in shapes:
    x = torch.randn(n, ci, h, w, device=dev).relu_()
    wt = torch.randn(co, ci, 3, 3, device=dev) * (2.0 / (9 * ci)) ** 0.5
    sx, sw = amax_of(x), amax_of(wt)
    wp = ops.conv3x3_pack(wt, sw)
    out = torch.empty(n, co, h, w, device=dev)
    ref = F.conv2d(x[:1].double(), wt.double(), padding=1)
    it = 3 if ci >= 512 else 30
    res = []
    for il in (0, 1):
        L.dcl_conv3x3_set_interleave(il)
        for _ in range(2):
            ops.conv3x3_launch(x, wp, co, sx, sw, out)
        torch.cuda.synchronize()
        err = ((out[:1].double() - ref).abs().max() / ref.abs().max()).item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            ops.conv3x3_launch(x, wp, co, sx, sw, out)
        e1.record()
        torch.cuda.synchronize()
        res.append((e0.elapsed_time(e1) / it * 1e3, err))
    print(f"  {n}x{ci}->{co} {h}x{w}: fenced {res[0][0]:8.1f} us (err {res[0][1]:.1e})  interleaved {res[1][0]:8.1f} us (err {res[1][1]:.1e})  "
          f"{res[0][0] / res[1][0]:.3f}x", flush=True)
    del x, wt, out
    torch.cuda.empty_cache()
L.dcl_conv3x3_set_interleave(2)
