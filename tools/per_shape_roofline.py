#!/usr/bin/env python3
"""Per-shape table of the direct f16x3 3x3 convolution kernels (forward = data gradient kernel, weight gradient) on
the convolution shapes of HRNet-W48 at the benchmark batch (12 x 512 x 1024 input): HIP-event time per launch,
algorithmic TFLOP/s and fraction of the f16x3 roofline (2.5 PFLOP/s / 3).  Writes CSV to stdout / --out so that
bench.py's `roofline.frac` can be recomputed from a tracked file (profiles/rNN_conv_per_shape.csv).

    python tools/per_shape_roofline.py --out gpurun_out/conv_per_shape.csv"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd  # noqa: E402,F401
from mscs_amd.models import ops  # noqa: E402
from mscs_amd.models.amax import amax_of  # noqa: E402

PEAK = 2500.0 / 3.0


def timeit(fn, iters):
    for _ in range(2):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--only", default=None, help="substring filter on the shape names")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    n = a.batch
    # (name, Cin, Cout, H, W, stride): BasicBlock convolutions of the four branches, the head, the stride-2 fuse convs
    shapes = [("branch0 basicblock", 48, 48, 128, 256, 1), ("branch1 basicblock", 96, 96, 64, 128, 1),
              ("branch2 basicblock", 192, 192, 32, 64, 1), ("branch3 basicblock", 384, 384, 16, 32, 1),
              ("cls_head (materialised concat, head_split off)", 720, 720, 128, 256, 1),
              ("cls_head fine part (x0 + up(x1))", 144, 720, 128, 256, 1), ("layer1 bottleneck", 64, 64, 128, 256, 1),
              ("fuse 48->96 s2", 48, 96, 128, 256, 2), ("fuse 96->192 s2", 96, 192, 64, 128, 2),
              ("fuse 192->384 s2", 192, 384, 32, 64, 2), ("fuse 48->48 s2", 48, 48, 128, 256, 2),
              ("stem conv2 s2", 64, 64, 256, 512, 2)]
    if a.only:
        shapes = [s for s in shapes if a.only in s[0]]
    rows = ["op,shape,n,cin,cout,h,w,stride,ms_per_launch,algorithmic_tflops,frac_of_f16x3_roofline"]
    gen = torch.Generator(device=dev).manual_seed(0)
    for name, ci, co, h, w, st in shapes:
        x = torch.randn(n, ci, h, w, device=dev, generator=gen).relu_()
        wt = torch.randn(co, ci, 3, 3, device=dev, generator=gen) * (2.0 / (9 * ci)) ** 0.5
        ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
        gy = torch.randn(n, co, ho, wo, device=dev, generator=gen) * 1e-3
        xa, wa, ga = amax_of(x), amax_of(wt), amax_of(gy)
        wp, wpt = ops.conv3x3_pack(wt, wa), ops.conv3x3_pack(wt, wa, True)
        y = torch.empty(n, co, ho, wo, device=dev)
        gx = torch.empty_like(x)
        flops = 2.0 * n * co * ci * 9 * ho * wo
        it = 5 if ci >= 720 else 30
        t_f = timeit(lambda: ops.conv3x3_launch(x, wp, co, xa, wa, y, stride=st), it)
        t_d = timeit(lambda: ops.conv3x3_launch(gy, wpt, ci, ga, wa, gx, in_up=st), it)
        t_w = timeit(lambda: ops.conv3x3_wgrad(x, gy, st), it) if ops.conv3x3_wgrad_supported(x, co) else float("nan")
        for op, t in (("forward", t_f), ("dgrad", t_d), ("wgrad(+reduce)", t_w)):
            tf = flops / (t * 1e-3) / 1e12
            rows.append(f"{op},{name},{n},{ci},{co},{h},{w},{st},{t:.4f},{tf:.1f},{tf / PEAK:.4f}")
        del x, wt, gy, y, gx
        torch.cuda.empty_cache()
    text = "\n".join(rows) + "\n"
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(text)


if __name__ == "__main__":
    main()
