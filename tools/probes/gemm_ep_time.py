#!/usr/bin/env python3
"""Per-launch time of the GEMM with and without its fused epilogues, beside the element-wise kernels they replace (Swin-L Mlp
shapes, batch 16 at 640 x 640)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd  # noqa
from mscs_amd.models import amax as am, ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for (M, C) in [(16 * 25600, 192), (16 * 6400, 384), (16 * 1600, 768), (16 * 400, 1536)]:
    Hd = 4 * C
    x = torch.randn(M, C, device=dev); w1 = torch.randn(Hd, C, device=dev) * C ** -0.5; b1 = torch.randn(Hd, device=dev)
    w2 = torch.randn(C, Hd, device=dev) * Hd ** -0.5; b2 = torch.randn(C, device=dev)
    h = torch.empty(M, Hd, device=dev); a = torch.empty(M, Hd, device=dev); y = torch.empty(M, C, device=dev); s = torch.randn(M, C, device=dev)
    gy = torch.randn(M, C, device=dev); gh = torch.empty(M, Hd, device=dev)
    ax, aw1, aw2, ag = am.amax_of(x), am.amax_of(w1), am.amax_of(w2), am.amax_of(gy)
    ca = am.zeros(1, dev)
    t = {}
    t["fc1"] = timeit(lambda: ops.gemm_f16x3(x, True, C, w1, True, C, M, Hd, C, h, Hd, ax, aw1, bias=b1, c_amax=ca))
    t["fc1+gelu ep"] = timeit(lambda: ops.gemm_f16x3_ep(x, w1, True, M, Hd, C, h, ax, aw1, 1, bias=b1, c_amax=ca, out2=a))
    t["gelu"] = timeit(lambda: torch.nn.functional.gelu(h))
    ah = am.amax_of(a)
    t["fc2"] = timeit(lambda: ops.gemm_f16x3(a, True, Hd, w2, True, Hd, M, C, Hd, y, C, ah, aw2, bias=b2, c_amax=ca))
    t["fc2+res ep"] = timeit(lambda: ops.gemm_f16x3_ep(a, w2, True, M, C, Hd, y, ah, aw2, 3, bias=b2, c_amax=ca, aux=s))
    t["add"] = timeit(lambda: torch.add(s, y))
    t["fc2 dgrad"] = timeit(lambda: ops.gemm_f16x3(gy, True, C, w2, False, Hd, M, Hd, C, gh, Hd, ag, aw2, c_amax=ca))
    t["fc2 dgrad+gelu' ep"] = timeit(lambda: ops.gemm_f16x3_ep(gy, w2, False, M, Hd, C, gh, ag, aw2, 2, c_amax=ca, aux=h))
    t["gelu_bwd"] = timeit(lambda: torch.ops.aten.gelu_backward(gh, h, approximate="none"))
    print(f"M {M} C {C}: " + ", ".join(f"{k} {v:.0f}" for k, v in t.items()), flush=True)
