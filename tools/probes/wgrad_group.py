"""Weight gradient: the four waves of a workgroup on adjacent strips (default) against four row ranges of one strip."""
import sys, torch
sys.path.insert(0, "/root/repo")
import mscs_amd
from mscs_amd import _lib
from mscs_amd.models import ops
L = _lib.lib()
dev = torch.device("cuda:0")
for shape in [(12, 48, 48, 128, 256), (12, 64, 64, 128, 256), (12, 96, 96, 64, 128), (12, 192, 192, 32, 64), (12, 384, 384, 16, 32),
              (16, 512, 512, 160, 160), (2, 48, 96, 19, 40), (3, 32, 64, 9, 72), (2, 64, 64, 33, 128)]:
    n, ci, co, h, w = shape
    x = torch.randn(n, ci, h, w, device=dev).relu_(); gy = torch.randn(n, co, h, w, device=dev) * 1e-3
    res, out = {0: [], 1: []}, {}
    for rep in range(3):
        for mode in (0, 1):
            L.dcl_wgrad3x3_set_strip_group(mode)
            for _ in range(2):
                out[mode] = ops.conv3x3_wgrad(x, gy)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            it = 3 if h >= 160 else 20
            e0.record()
            for _ in range(it):
                ops.conv3x3_wgrad(x, gy)
            e1.record(); torch.cuda.synchronize()
            res[mode].append(round(e0.elapsed_time(e1) / it * 1e3, 1))
    d = ((out[0] - out[1]).abs().max() / out[0].abs().max()).item()
    print(shape, "rows-of-one-strip us", res[0], "adjacent strips us", res[1], "rel diff", d)
    del x, gy
    torch.cuda.empty_cache()
L.dcl_wgrad3x3_set_strip_group(1)
