"""Bilinear up-sampling backward (k_upsample_bwd_rows) through the C ABI on the step's shapes: time per launch and a checksum
(DCL_LIB_PATH selects the build: bitwise comparison between builds by the checksums)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd  # noqa: E402,F401
from mscs_amd.models import ops  # noqa: E402

dev = torch.device("cuda:0")
print("library:", os.environ.get("DCL_LIB_PATH", "product"))
for (c, h, w, H, W) in [(48, 64, 128, 128, 256), (48, 32, 64, 128, 256), (48, 16, 32, 128, 256), (96, 32, 64, 64, 128), (7, 9, 12, 37, 52), (5, 10, 16, 20, 32)]:
    torch.manual_seed(c + h)
    x = torch.randn(12 if c > 8 else 2, c, h, w, device=dev, requires_grad=True)
    y = ops.upsample_bilinear(x, (H, W), True)
    gy = torch.randn_like(y)
    from mscs_amd import _lib
    L = _lib.lib()
    gx = torch.empty_like(x)
    planes = x.shape[0] * c
    st = _lib.stream_ptr(dev)
    call = lambda: _lib.check(L.dcl_upsample_bilinear_bwd(_lib.ptr(gy), planes, h, w, H, W, 1, _lib.ptr(gx), st), "bwd")
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        call()
    e1.record()
    torch.cuda.synchronize()
    xr = x.detach().double().requires_grad_(True)
    yr = torch.nn.functional.interpolate(xr, size=(H, W), mode="bilinear", align_corners=True)
    ref = torch.autograd.grad(yr, xr, gy.double())[0]
    err = ((gx.double() - ref).abs().max() / ref.abs().max()).item()
    print(f"  {x.shape[0]}x{c}x{h}x{w} <- {H}x{W}: {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us   distance to float64 {err:.1e}   "
          f"checksum {gx.view(torch.int32).sum(dtype=torch.int64).item()}")
