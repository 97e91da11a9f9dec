"""UPerNet decoder weight gradient (8 x 512 -> 512 x 160 x 160): tile / variant / split options."""
import sys, torch
sys.path.insert(0, "/root/repo")
import mscs_amd
from mscs_amd import _lib
from mscs_amd.models import ops
L = _lib.lib()
dev = torch.device("cuda:0")
shape = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (8, 512, 512, 160, 160)
n, ci, co, h, w = shape
x = torch.randn(n, ci, h, w, device=dev).relu_(); gy = torch.randn(n, co, h, w, device=dev) * 1e-3
ref = None
for (variant, tile, nx) in [(-1, (0, 0), 0), (-1, (2, 1), 0), (-1, (1, 2), 0), (-1, (1, 1), 0), (0, (0, 0), 0), (-1, (0, 0), 2), (-1, (2, 1), 1)]:
    L.dcl_wgrad3x3_set_variant(variant); L.dcl_wgrad3x3_set_tile(*tile); L.dcl_wgrad3x3_set_splits(nx)
    try:
        for _ in range(2):
            gw = ops.conv3x3_wgrad(x, gy)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            gw = ops.conv3x3_wgrad(x, gy)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        if ref is None:
            ref = gw
        fl = 2 * 9 * ci * co * n * h * w
        print("variant", variant, "tile", tile, "nx", nx, "slabs", L.dcl_wgrad3x3_splits(n, ci, co, h, w, 1), "ms", round(ms, 3),
              "frac", round(fl / (ms * 1e-3) / 833.3e12, 3), "diff", ((gw - ref).abs().max() / ref.abs().max()).item())
    except Exception as e:
        print("variant", variant, "tile", tile, "nx", nx, "ERR", e)
L.dcl_wgrad3x3_set_variant(-1); L.dcl_wgrad3x3_set_tile(0, 0); L.dcl_wgrad3x3_set_splits(0)
