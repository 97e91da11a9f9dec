"""HIP-event times of the two per-plane reduction kernels of the fused norm (statistics, backward reduce) on the step's shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd  # noqa: E402,F401
from mscs_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
st = _lib.stream_ptr(dev)
print("library:", os.environ.get("DCL_LIB_PATH", "product"))
for (n, c, h, w) in [(12, 48, 128, 256), (12, 96, 64, 128), (12, 192, 32, 64), (12, 384, 16, 32), (12, 256, 128, 256), (12, 64, 128, 256)]:
    hw = h * w
    x = torch.randn(n, c, h, w, device=dev)
    dy = torch.randn(n, c, h, w, device=dev)
    ns = L.dcl_bn_num_slices(n, c)
    part = torch.empty(c * ns * 2, device=dev)
    rm = torch.zeros(c, device=dev)
    piv = torch.empty(c, device=dev)
    mean = torch.zeros(c, device=dev)
    inv = torch.ones(c, device=dev)
    g = torch.ones(c, device=dev)
    b = torch.zeros(c, device=dev)

    def stats():
        _lib.check(L.dcl_bn_stats_part(_lib.ptr(x), n, c, hw, _lib.ptr(part), _lib.ptr(rm), _lib.ptr(piv), st), "stats")

    def reduce_():
        _lib.check(L.dcl_bn_bwd_reduce_part(_lib.ptr(dy), _lib.ptr(x), None, _lib.ptr(mean), _lib.ptr(inv), _lib.ptr(g), _lib.ptr(b), n, c, hw, 1,
                                            _lib.ptr(part), st), "reduce")

    out = []
    for fn, passes in ((stats, 1), (reduce_, 2)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        out.append(f"{us:7.1f} us ({passes * x.numel() * 4 / us / 1e6:5.2f} TB/s)")
    print(f"  12x{c}x{h}x{w} (ns {ns}): statistics {out[0]}   backward reduce {out[1]}")
