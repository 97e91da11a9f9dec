"""Weight gradient: pixel splits per tile pair (dcl_wgrad3x3_set_splits) on the UPerNet decoder / HRNet shapes."""
import sys, torch
sys.path.insert(0, "/root/repo")
import mscs_amd
from mscs_amd import _lib
from mscs_amd.models import ops
L = _lib.lib()
dev = torch.device("cuda:0")
for shape, nxs in [((16, 512, 512, 160, 160), (0, 2, 3, 4)), ((16, 1024, 512, 160, 160), (0, 2)), ((16, 512, 512, 80, 80), (0, 2, 4)),
                   ((16, 512, 512, 40, 40), (0, 2, 4)), ((12, 384, 384, 16, 32), (0, 1, 2)), ((12, 192, 192, 32, 64), (0, 6, 8, 10)),
                   ((12, 96, 96, 64, 128), (0, 24, 32, 42)), ((12, 48, 48, 128, 256), (0, 128, 170))]:
    n, ci, co, h, w = shape
    x = torch.randn(n, ci, h, w, device=dev).relu_(); gy = torch.randn(n, co, h, w, device=dev) * 1e-3
    out = []
    for nx in nxs:
        L.dcl_wgrad3x3_set_splits(nx)
        for _ in range(2):
            ops.conv3x3_wgrad(x, gy)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 3 if h >= 160 else 20
        e0.record()
        for _ in range(it):
            ops.conv3x3_wgrad(x, gy)
        e1.record(); torch.cuda.synchronize()
        out.append((nx, L.dcl_wgrad3x3_splits(n, ci, co, h, w, 1), round(e0.elapsed_time(e1) / it * 1e3, 1)))
    print(shape, "(nx, slabs, us):", out)
    del x, gy
    torch.cuda.empty_cache()
L.dcl_wgrad3x3_set_splits(0)
