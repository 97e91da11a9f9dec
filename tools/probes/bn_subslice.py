#!/usr/bin/env python3
"""Would the BN reduction kernels gain from slices finer than one plane?  A plane (n, c) of HW values cut into `sub` pieces is the
tensor [N, sub C, HW / sub] with c' = sub c + piece: the existing entry points, called with those extents, walk the memory
exactly as a sub-sliced launch would (grid (sub C, min(N, 2048 / (sub C)))).  HIP-event time per launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd  # noqa
from mscs_amd import _lib
L = _lib.lib(); p = _lib.ptr
dev = torch.device("cuda:0")
st = _lib.stream_ptr(dev)
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
N = 12
for (C, H, W) in [(48, 128, 256), (96, 64, 128), (192, 32, 64), (384, 16, 32), (64, 256, 512), (256, 128, 256), (720, 128, 256)]:
    HW = H * W
    x = torch.randn(N, C, HW, device=dev); dy = torch.randn(N, C, HW, device=dev)
    mb = x.numel() * 4 / 1e6
    line = f"{C:3d} ch {H}x{W} ({mb:5.0f} MB):"
    for sub in (1, 2, 4, 8):
        if HW % (256 * 4 * sub) or C * sub > 8192: continue
        Cs, HWs = C * sub, HW // sub
        ns = L.dcl_bn_num_slices(N, Cs)
        part = torch.empty(Cs * ns * 2, device=dev)
        mean = torch.zeros(Cs, device=dev); inv = torch.ones(Cs, device=dev); g = torch.ones(Cs, device=dev); b = torch.zeros(Cs, device=dev)
        t1 = timeit(lambda: L.dcl_bn_stats_part(p(x), N, Cs, HWs, p(part), None, None, st))
        t2 = timeit(lambda: L.dcl_bn_bwd_reduce_part(p(dy), p(x), None, p(mean), p(inv), p(g), p(b), N, Cs, HWs, 1, p(part), st))
        line += f"  sub {sub} (grid {Cs}x{ns}): stats {t1:5.1f} us {mb / t1 / 1e0 / 1e3:4.2f} TB/s, reduce {t2:5.1f} us {2 * mb / t2 / 1e3:4.2f} TB/s;"
    print(line, flush=True)
