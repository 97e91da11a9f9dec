#!/usr/bin/env python3
"""LayerNorm forward / backward per launch on the Swin shapes, with and without the absmax side channel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd  # noqa
from mscs_amd import _lib
from mscs_amd.models import amax as am
L = _lib.lib(); p = _lib.ptr
dev = torch.device("cuda:0")
st = _lib.stream_ptr(dev)
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for (M, C) in [(16 * 25600, 192), (16 * 6400, 384), (16 * 1600, 768), (16 * 400, 1536), (16 * 16384, 96)]:
    x = torch.randn(M, C, device=dev); gy = torch.randn(M, C, device=dev); w = torch.ones(C, device=dev); b = torch.zeros(C, device=dev)
    y = torch.empty_like(x); gx = torch.empty_like(x); stats = torch.empty(2, M, device=dev)
    tag = am.zeros(am.SLOTS, dev)
    parts = torch.empty((L.dcl_layernorm_bwd_parts(M, C), 2, C), device=dev); gwb = torch.empty(2, C, device=dev)
    t = {}
    t["fwd+tag"] = timeit(lambda: L.dcl_layernorm_fwd(p(x), p(w), p(b), M, C, 1e-5, p(y), p(stats[0]), p(stats[1]), p(tag), st))
    t["fwd"] = timeit(lambda: L.dcl_layernorm_fwd(p(x), p(w), p(b), M, C, 1e-5, p(y), p(stats[0]), p(stats[1]), None, st))
    t["bwd+tag"] = timeit(lambda: L.dcl_layernorm_bwd(p(gy), p(x), p(w), p(stats[0]), p(stats[1]), M, C, p(gx), p(parts), p(gwb), None, p(tag), st))
    t["bwd"] = timeit(lambda: L.dcl_layernorm_bwd(p(gy), p(x), p(w), p(stats[0]), p(stats[1]), M, C, p(gx), p(parts), p(gwb), None, None, st))
    mb = x.numel() * 4 / 1e6
    print(f"M {M} C {C} ({mb:.0f} MB): " + ", ".join(f"{k} {v:.0f} us" for k, v in t.items()), flush=True)
