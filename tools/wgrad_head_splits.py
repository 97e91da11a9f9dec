import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import mscs_amd
from mscs_amd import _lib
from mscs_amd.models import ops
from per_shape_roofline import timeit
L = _lib.lib(); dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(12, 720, 128, 256, device=dev, generator=gen).relu_()
gy = torch.randn(12, 720, 128, 256, device=dev, generator=gen) * 1e-3
ref = ops.conv3x3_wgrad(x, gy)
for nx in (0, 1, 2, 3, 4):
    L.dcl_wgrad3x3_set_splits(nx)
    out = ops.conv3x3_wgrad(x, gy)
    t = timeit(lambda: ops.conv3x3_wgrad(x, gy), 8)
    print(f"nx={nx}: {t*1e3:8.1f} us  diff {((out-ref).abs().max()/ref.abs().max()).item():.1e}", flush=True)
