"""Wave-form weight gradient (head 144 -> 720): rows per column of the traversal."""
import sys, torch
sys.path.insert(0, "/root/repo")
import mscs_amd
from mscs_amd import _lib
from mscs_amd.models import ops
L = _lib.lib()
dev = torch.device("cuda:0")
for shape in [(12, 144, 720, 128, 256), (12, 384, 384, 16, 32), (2, 144, 720, 64, 40)]:
    n, ci, co, h, w = shape
    x = torch.randn(n, ci, h, w, device=dev).relu_(); gy = torch.randn(n, co, h, w, device=dev) * 1e-3
    ref = None
    for band in (0, 64, 32, 16, 8, 0, 32):
        L.dcl_wgrad3x3_set_wave_band(band)
        for _ in range(2):
            gw = ops.conv3x3_wgrad(x, gy)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv3x3_wgrad(x, gy)
        e1.record(); torch.cuda.synchronize()
        ref = gw if ref is None else ref
        print(shape, "band", band, "us", round(e0.elapsed_time(e1) * 100, 1), "diff", ((gw - ref).abs().max() / ref.abs().max()).item())
L.dcl_wgrad3x3_set_wave_band(32)
