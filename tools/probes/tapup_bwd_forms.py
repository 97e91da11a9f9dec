"""Tap-up backward: first form against the windowed form -- bitwise-near agreement, fp64 check of both on a small case, timing."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd
from mscs_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
st = _lib.stream_ptr(dev)


def run(dy, h, w, align, form, cm=1):
    n, co, H, W = dy.shape
    L.dcl_tapup_set_bwd_form(form)
    dz = torch.full((9 * co, n * h * w), float("nan"), device=dev)
    _lib.check(L.dcl_tapup_bwd(_lib.ptr(dy), n, co, H, W, h, w, align, cm, _lib.ptr(dz), st), "tapup_bwd")
    return dz


for (n, co, H, W, h, w) in [(2, 5, 32, 64, 8, 16), (1, 3, 33, 40, 9, 7), (2, 4, 20, 20, 10, 10), (1, 2, 19, 300, 5, 38),
                            (1, 3, 16, 24, 16, 24), (1, 2, 40, 40, 3, 3), (1, 2, 64, 64, 2, 2), (2, 32, 24, 40, 6, 10), (2, 32, 24, 40, 3, 5),
                            (1, 2, 24, 24, 1, 1), (1, 2, 9, 8, 2, 3)]:
    for align in (1, 0):
        dy = torch.randn(n, co, H, W, device=dev)
        a, b = run(dy, h, w, align, 1), run(dy, h, w, align, 2)
        print((n, co, H, W, h, w), "align", align, "max |form2 - form1|", (a - b).abs().max().item(), "max", a.abs().max().item())
for (n, co, H, W, h, w) in [(12, 720, 128, 256, 32, 64), (12, 720, 128, 256, 16, 32), (8, 512, 160, 160, 40, 40),
                            (8, 512, 160, 160, 80, 80), (8, 512, 160, 160, 20, 20)]:
    dy = torch.randn(n, co, H, W, device=dev)
    res = {}
    for form in (1, 2, 1, 2):
        run(dy, h, w, 1, form)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dz = torch.empty((9 * co, n * h * w), device=dev)
        e0.record()
        for _ in range(5):
            L.dcl_tapup_bwd(_lib.ptr(dy), n, co, H, W, h, w, 1, 1, _lib.ptr(dz), st)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(form, []).append(round(e0.elapsed_time(e1) / 5, 3))
    gb = (dy.numel() + 9 * co * n * h * w) * 4 / 1e9
    print((n, co, H, W, h, w), "ms form1", res[1], "form2", res[2], "algorithmic GB", round(gb, 2))
# planes per workgroup of the windowed form (dcl_tapup_set_bwd_form(16 + p))
L.dcl_tapup_set_bwd_form(2)
for (n, co, H, W, h, w) in [(12, 720, 128, 256, 32, 64), (12, 720, 128, 256, 16, 32), (16, 512, 128, 128, 32, 32), (16, 512, 128, 128, 16, 16),
                            (16, 512, 160, 160, 40, 40), (16, 512, 160, 160, 20, 20)]:
    dy = torch.randn(n, co, H, W, device=dev)
    out = []
    ref = None
    for p in (1, 2, 4, 8, 1, 2, 4, 8):
        L.dcl_tapup_set_bwd_form(16 + p)
        dz = torch.empty((9 * co, n * h * w), device=dev)
        L.dcl_tapup_bwd(_lib.ptr(dy), n, co, H, W, h, w, 0, 1, _lib.ptr(dz), st)
        torch.cuda.synchronize()
        if ref is None:
            ref = dz.clone()
        assert torch.equal(ref, dz), "planes per workgroup changed the result"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            L.dcl_tapup_bwd(_lib.ptr(dy), n, co, H, W, h, w, 0, 1, _lib.ptr(dz), st)
        e1.record(); torch.cuda.synchronize()
        out.append((p, round(e0.elapsed_time(e1) / 5, 3)))
    print((n, co, H, W, h, w), "ms by planes per workgroup", out, "dy GB", round(dy.numel() * 4 / 1e9, 2))
L.dcl_tapup_set_bwd_form(16 + 4)
