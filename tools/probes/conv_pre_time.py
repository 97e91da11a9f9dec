"""Stand-alone times of the deferred-norm pieces against the pieces they replace, on the BasicBlock shapes of HRNet-W48 at batch 12
(C ABI, HIP events, alternating):   gpurun -- python tools/probes/conv_pre_time.py
  norm:  stats + apply(relu)            vs  stats_minmax + finalize_pre, or dcl_bn_stats_pre (one launch)
  conv2: dcl_conv3x3_f16x3 on y         vs  dcl_conv3x3_pre_f16x3 on z
  wgrad: dcl_wgrad3x3_f16x3 on y        vs  dcl_wgrad3x3_pre_f16x3 on z"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd  # noqa: F401,E402
from mscs_amd import _lib as P  # noqa: E402
from mscs_amd.models import amax as A, ops  # noqa: E402


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    L = P.lib()
    N = 12
    for (c, h, w) in [(48, 128, 256), (96, 64, 128), (192, 32, 64), (384, 16, 32)]:
        torch.manual_seed(c)
        z = torch.randn(N, c, h, w, device=dev)
        gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        nbt = torch.zeros(1, dtype=torch.int64, device=dev)
        ns = L.dcl_bn_num_slices(N, c)
        part, mm = torch.empty(c * ns * 2, device=dev), torch.empty(c * ns * 2, device=dev)
        mean, invstd, pivot, sc, sh = (torch.empty(c, device=dev) for _ in range(5))
        y = torch.empty_like(z)
        amax_w, amax_d = torch.zeros(A.SLOTS, device=dev), torch.zeros(A.SLOTS, device=dev)
        st = P.stream_ptr(dev)
        cnt = float(N * h * w)

        def stats():
            L.dcl_bn_stats_part(P.ptr(z), N, c, h * w, P.ptr(part), P.ptr(rm), P.ptr(pivot), st)

        def apply():
            L.dcl_bn_apply_parts(P.ptr(z), None, P.ptr(part), ns, cnt, 1e-5, 0.0, P.ptr(gamma), P.ptr(beta), N, c, h * w, 1, P.ptr(y),
                                 P.ptr(mean), P.ptr(invstd), P.ptr(rm), P.ptr(rv), P.ptr(nbt), P.ptr(amax_w), P.ptr(pivot), None, st)

        def stats_mm():
            L.dcl_bn_stats_minmax_part(P.ptr(z), N, c, h * w, P.ptr(part), P.ptr(mm), P.ptr(rm), P.ptr(pivot), st)

        tickets = torch.zeros(c, dtype=torch.int32, device=dev)

        def stats_pre():
            tickets.zero_()
            L.dcl_bn_stats_pre(P.ptr(z), N, c, h * w, P.ptr(part), P.ptr(mm), P.ptr(tickets), cnt, 1e-5, 0.0, P.ptr(gamma), P.ptr(beta),
                               P.ptr(mean), P.ptr(invstd), P.ptr(rm), P.ptr(rv), P.ptr(nbt), P.ptr(sc), P.ptr(sh), P.ptr(amax_d), st)

        def finalize():
            L.dcl_bn_finalize_pre(P.ptr(part), P.ptr(mm), ns, cnt, 1e-5, 0.0, P.ptr(gamma), P.ptr(beta), c, P.ptr(mean), P.ptr(invstd),
                                  P.ptr(rm), P.ptr(rv), P.ptr(nbt), P.ptr(pivot), P.ptr(sc), P.ptr(sh), P.ptr(amax_d), st)

        stats(); apply(); stats_mm(); finalize()
        wt = torch.randn(c, c, 3, 3, device=dev) * (2.0 / (9 * c)) ** 0.5
        wamax = A.amax_of(wt)
        wp = ops.conv3x3_pack(wt, wamax)
        out = torch.empty_like(z)
        gy = torch.randn(N, c, h, w, device=dev) * 1e-3
        ga = A.amax_of(gy)
        splits = L.dcl_wgrad3x3_splits(N, c, c, h, w, 1)
        wpart = torch.empty(splits * 9 * c * c, device=dev)
        dw = torch.empty(c, c, 3, 3, device=dev)

        def conv():
            L.dcl_conv3x3_f16x3(P.ptr(y), N, c, h, w, P.ptr(wp), c, P.ptr(amax_w), A.SLOTS, P.ptr(wamax), None, None, P.ptr(out), 1, 1,
                                h, w, 0, 0, st)

        def conv_pre():
            L.dcl_conv3x3_pre_f16x3(P.ptr(z), N, c, h, w, P.ptr(wp), c, P.ptr(amax_d), A.SLOTS, P.ptr(wamax), P.ptr(sc), P.ptr(sh), None,
                                    P.ptr(out), 1, 0, 0, st)

        def wgrad():
            L.dcl_wgrad3x3_f16x3(P.ptr(y), P.ptr(gy), N, c, c, h, w, P.ptr(amax_w), A.SLOTS, P.ptr(ga), ga.numel(), 1, P.ptr(wpart),
                                 P.ptr(dw), st)

        def wgrad_pre():
            L.dcl_wgrad3x3_pre_f16x3(P.ptr(z), P.ptr(gy), N, c, c, h, w, P.ptr(amax_d), A.SLOTS, P.ptr(ga), ga.numel(), P.ptr(sc),
                                     P.ptr(sh), 1, P.ptr(wpart), P.ptr(dw), st)

        res = {}
        for rnd in range(2):
            for name, fn in (("stats", stats), ("stats_mm", stats_mm), ("stats_pre(+4us memset)", stats_pre), ("apply", apply),
                             ("finalize", finalize), ("conv", conv),
                             ("conv_pre", conv_pre), ("wgrad", wgrad), ("wgrad_pre", wgrad_pre)):
                res.setdefault(name, []).append(timeit(fn))
        print(f"{c:4d} ch {h}x{w}: " + "  ".join(f"{k} {min(v):.1f}" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
