#!/usr/bin/env python3
"""dcl_gemm_f16x3 against the library's fp32 GEMM on the shapes the models run: error vs float64 and time per launch.

    python tools/gemm_shapes.py [--tile T] [--only swinl|swint|head|conv1x1] [--check]

Rows: the three GEMMs of a Linear (forward x W^T, data gradient dy W, weight gradient dy^T x) per Swin stage and kind,
the head's tap products, HRNet's large 1x1 convolutions (batched over the images).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: F401,E402  (registers the package alias)
from mscs_amd import _lib  # noqa: E402
from mscs_amd.models import ops  # noqa: E402
from mscs_amd.models.amax import amax_of  # noqa: E402


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def linear_rows(name, m, k, n):
    """(label, f16x3 callable, library callable, flop, float64 reference maker)"""
    dev = "cuda"
    x = torch.randn(m, k, device=dev)
    w = torch.randn(n, k, device=dev) * 0.05
    gy = torch.randn(m, n, device=dev) * 1e-3
    for t in (x, w, gy):
        amax_of(t)
    flop = 2.0 * m * n * k
    return [
        (f"{name} fwd   [{m}x{k}] . [{n}x{k}]^T", lambda: ops.linear_f16x3(x, w, None, tag_out=False),
         lambda: torch.nn.functional.linear(x, w), flop, lambda: x.double() @ w.double().t()),
        (f"{name} dgrad [{m}x{n}] . [{n}x{k}]", lambda: ops.linear_dgrad_f16x3(gy, w), lambda: gy.mm(w), flop,
         lambda: gy.double() @ w.double()),
        (f"{name} wgrad [{m}x{n}]^T . [{m}x{k}]", lambda: ops.linear_wgrad_f16x3(gy, x), lambda: gy.t().mm(x), flop,
         lambda: gy.double().t() @ x.double()),
    ]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--only", default="")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--shape", default="", help="m,k,n: one Linear only")
    ap.add_argument("--which", default="", help="fwd|dgrad|wgrad: only that GEMM of each Linear")
    ap.add_argument("--no-library", action="store_true")
    ap.add_argument("--sweep-tiles", action="store_true", help="time every forced tile (1..5) next to the automatic plan")
    args = ap.parse_args()
    L = _lib.lib()
    L.dcl_gemm_set_tile(args.tile)
    torch.manual_seed(0)
    rows = []
    if args.shape:
        m, k, n = (int(v) for v in args.shape.split(","))
        rows.append(("shape", m, k, n))
        args.only = "none"
    if args.only in ("", "swinl"):
        # Swin-L 640^2 batch 16: tokens per stage, C = 192 * 2^s
        for s, m in enumerate((409600, 102400, 25600, 6400)):
            c = 192 << s
            for kind, k, n in (("qkv", c, 3 * c), ("proj", c, c), ("fc1", c, 4 * c), ("fc2", 4 * c, c)):
                rows.append((f"swinL s{s + 1} {kind}", m, k, n))
    if args.only in ("", "swint"):
        for s, m in enumerate((262144, 65536, 16384, 4096)):
            c = 96 << s
            for kind, k, n in (("qkv", c, 3 * c), ("fc2", 4 * c, c)):
                rows.append((f"swinT s{s + 1} {kind}", m, k, n))
    head_rows = []
    if args.only in ("", "head"):
        # the head's tap products (models/ops.py _CoarseTaps): z = W_b x_b, dx_b = W_b^T dz, dW_b = dz x_b^T
        for cb, p in ((192, 12 * 32 * 64), (384, 12 * 16 * 32)):          # HRNet-W48: the 1/4- and 1/8-resolution branches
            head_rows.append((cb, p))
    print("shape,ms_f16x3,tflops_f16x3,frac_of_833,ms_library_f32,speedup,err_f16x3,err_library")
    groups = [linear_rows(name, m, k, n) for name, m, k, n in rows] if not head_rows else []
    for name, m, k, n in (rows if head_rows else []):
        groups.append(linear_rows(name, m, k, n))
    for cb, P in head_rows:
        groups.append(head_gemms(cb, P))
    for group in groups:
        m = n = 0
        for label, mine, lib, flop, ref in group:
            if args.which and f" {args.which} " not in label.replace("  ", " "):
                continue
            tm = timeit(mine, args.iters)
            if args.sweep_tiles:
                ts = []
                for t in range(1, 6):
                    L.dcl_gemm_set_tile(t)
                    try:
                        ts.append(timeit(mine, args.iters))
                    except RuntimeError:
                        ts.append(float("nan"))
                L.dcl_gemm_set_tile(args.tile)
                best = min(range(5), key=lambda i: ts[i] if ts[i] == ts[i] else 1e9)
                print(f"{label},auto {tm:.4f},best tile {best + 1} {ts[best]:.4f},ratio {tm / ts[best]:.2f}," +
                      " ".join(f"{v:.4f}" for v in ts), flush=True)
                continue
            tl = float("nan") if args.no_library else timeit(lib, args.iters)
            e1 = e2 = float("nan")
            if not args.no_library and (args.check or flop <= 4e11):
                r = ref()
                sc = r.abs().max()
                e1 = ((mine().double() - r).abs().max() / sc).item()
                e2 = ((lib().double() - r).abs().max() / sc).item()
                del r
            tf = flop / tm / 1e9
            print(f"{label},{tm:.4f},{tf:.1f},{tf / 833.3:.3f},{tl:.4f},{tl / tm:.2f},{e1:.2e},{e2:.2e}", flush=True)
        torch.cuda.empty_cache()


def head_gemms(cb, P, co9=6480):
    dev = "cuda"
    wb = torch.randn(co9, cb, device=dev) * 0.05
    xc = torch.randn(cb, P, device=dev)
    dz = torch.randn(co9, P, device=dev) * 1e-3
    am = {id(t): amax_of(t) for t in (wb, xc, dz)}
    z, gx, gw = torch.empty(co9, P, device=dev), torch.empty(cb, P, device=dev), torch.empty(co9, cb, device=dev)
    flop = 2.0 * co9 * cb * P
    return [
        (f"head cb={cb} z = W x   [{co9}x{cb}] . [{cb}x{P}]",
         lambda: ops.gemm_f16x3(wb, True, cb, xc, False, P, co9, P, cb, z, P, am[id(wb)], am[id(xc)], splitk=1),
         lambda: torch.mm(wb, xc), flop, lambda: wb.double() @ xc.double()),
        (f"head cb={cb} dx = W^T dz [{cb}x{co9}] . [{co9}x{P}]",
         lambda: ops.gemm_f16x3(wb, False, cb, dz, False, P, cb, P, co9, gx, P, am[id(wb)], am[id(dz)], splitk=1),
         lambda: torch.mm(wb.t(), dz), flop, lambda: wb.double().t() @ dz.double()),
        (f"head cb={cb} dW = dz x^T [{co9}x{P}] . [{P}x{cb}]",
         lambda: ops.gemm_f16x3(dz, True, P, xc, True, P, co9, cb, P, gw, cb, am[id(dz)], am[id(xc)]),
         lambda: torch.mm(dz, xc.t()), flop, lambda: dz.double() @ xc.double().t()),
    ]


if __name__ == "__main__":
    main()
