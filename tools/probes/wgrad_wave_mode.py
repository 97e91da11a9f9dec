"""Weight gradient with 129 .. 256 tile pairs: one workgroup per pair against wave-level pixel splits (alternating)."""
import sys, torch
sys.path.insert(0, "/root/repo")
import mscs_amd
from mscs_amd import _lib
from mscs_amd.models import ops
L = _lib.lib()
dev = torch.device("cuda:0")
shapes = [(12, 144, 720, 128, 256), (12, 384, 384, 16, 32), (12, 192, 192, 32, 64)]
for (n, ci, co, h, w) in shapes:
    x = torch.randn(n, ci, h, w, device=dev).relu_(); gy = torch.randn(n, co, h, w, device=dev) * 1e-3
    res = {0: [], 1: [], 2: []}
    for rep in range(4):
        for mode in (0, 1, 2):
            L.dcl_wgrad3x3_set_wave_mode(mode)
            for _ in range(3):
                ops.conv3x3_wgrad(x, gy)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv3x3_wgrad(x, gy)
            e1.record(); torch.cuda.synchronize()
            res[mode].append(e0.elapsed_time(e1) / 10 * 1e3)
    L.dcl_wgrad3x3_set_wave_mode(2)
    fl = 2 * 9 * ci * co * n * h * w
    print((n, ci, co, h, w), "slabs", L.dcl_wgrad3x3_splits(n, ci, co, h, w, 1),
          "workgroup/pair us", [round(v, 1) for v in res[0]], "wave splits us", [round(v, 1) for v in res[1]], "pairs-of-a-split us", [round(v, 1) for v in res[2]],
          "frac", round(fl / (min(res[2]) * 1e-6) / 833.3e12, 3))
