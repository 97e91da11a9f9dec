#!/usr/bin/env python3
"""Direct f16x3 3x3 convolution (csrc/dcl_conv3x3.hip) against MIOpen f32 on HRNet-W48's BasicBlock shapes:
accuracy against an fp64 convolution and time per call.   python tools/bench_conv3x3.py [--tiles]"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd  # noqa: E402
from mscs_amd.models import ops  # noqa: E402
from mscs_amd.models.amax import amax_of  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    shapes = [(12, 48, 128, 256), (12, 96, 64, 128), (12, 192, 32, 64), (12, 384, 16, 32)]
    sweep = "--tiles" in sys.argv
    for (n, c, h, w) in shapes:
        x = torch.randn(n, c, h, w, device=dev).relu_()
        wt = torch.randn(c, c, 3, 3, device=dev) * (2.0 / (9 * c)) ** 0.5
        ref64 = F.conv2d(x[:2].double(), wt.double(), padding=1)
        y = ops.conv3x3_direct(x, wt)
        y32 = F.conv2d(x, wt, padding=1)
        den = ref64.abs().max()
        e_dir = ((y[:2].double() - ref64).abs().max() / den).item()
        e_f32 = ((y32[:2].double() - ref64).abs().max() / den).item()
        flops = 2.0 * n * c * c * 9 * h * w
        t_mi = timeit(lambda: F.conv2d(x, wt, padding=1))
        t_all = timeit(lambda: ops.conv3x3_direct(x, wt))
        sx, sw = amax_of(x), amax_of(wt)
        wp = ops.conv3x3_pack(wt, sw)
        out = torch.empty_like(x)
        t_k = timeit(lambda: ops.conv3x3_launch(x, wp, c, sx, sw, out))
        line = (f"C={c:3d} {h}x{w}: err direct {e_dir:.2e} miopen-f32 {e_f32:.2e} | miopen {t_mi:.3f} ms "
                f"({flops / t_mi / 1e9:.0f} TF) direct {t_all:.3f} ms kernel {t_k:.3f} ms ({flops / t_k / 1e9:.0f} TF)")
        print(line, flush=True)
        if sweep:
            for r in (1, 2, 3):
                for p in (1, 2, 4):
                    t = timeit(lambda: ops.conv3x3_launch(x, wp, c, sx, sw, out, r, p), 10)
                    print(f"    R={r} P={p}: {t:.3f} ms ({flops / t / 1e9:.0f} TF)", flush=True)
        # data gradient through the same kernel
        gy = torch.randn_like(x) * 1e-4
        gx = ops.conv3x3_direct(gy, wt, transposed=True)
        gref = F.conv_transpose2d(gy[:2].double(), wt.double(), padding=1)
        print(f"    dgrad err {((gx[:2].double() - gref).abs().max() / gref.abs().max()).item():.2e}", flush=True)
        # weight gradient
        gw = ops.conv3x3_wgrad(x, gy)
        gw_lib = torch.ops.aten.convolution_backward(gy, x, wt, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                     [False, True, False])[1]
        gw64 = torch.ops.aten.convolution_backward(gy.double().cpu(), x.double().cpu(), wt.double().cpu(), None,
                                                   [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                   [False, True, False])[1].to(dev)
        dn = gw64.abs().max()
        t_w = timeit(lambda: ops.conv3x3_wgrad(x, gy))
        t_wl = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, [1, 1], [1, 1], [1, 1], False,
                                                                  [0, 0], 1, [False, True, False]))
        print(f"    wgrad err direct {((gw.double() - gw64).abs().max() / dn).item():.2e} miopen "
              f"{((gw_lib.double() - gw64).abs().max() / dn).item():.2e} | miopen {t_wl:.3f} ms direct {t_w:.3f} ms "
              f"({flops / t_w / 1e9:.0f} TF)", flush=True)


if __name__ == "__main__":
    main()
