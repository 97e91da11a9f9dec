import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd
from mscs_amd.models import ops
from mscs_amd.models.amax import amax_of
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for ci in (16, 48, 96, 144, 288):
    co, n, h, w = 48, 12, 128, 256
    x = torch.randn(n, ci, h, w, device=dev).relu_(); wt = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    sx, sw = amax_of(x), amax_of(wt)
    wp = ops.conv3x3_pack(wt, sw)
    out = torch.empty(n, co, h, w, device=dev)
    for (r, p) in ((2, 4), (2, 2), (2, 1)):
        t = timeit(lambda: ops.conv3x3_launch(x, wp, co, sx, sw, out, r, p))
        print(f"Cin={ci:3d} Cout=48 ({r},{p}): {t*1e3:.1f} us", flush=True)
