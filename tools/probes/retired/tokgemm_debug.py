import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd
from mscs_amd.models import ops
from mscs_amd.models.amax import amax_of
dev = torch.device("cuda:0")
torch.manual_seed(0)
def split(t, amax):
    s = 2.0 ** torch.floor(torch.log2(16384.0 / amax))
    ts = t.double() * s
    hi = ts.to(torch.float16).double()
    lo = (ts - hi).to(torch.float16).double()
    return hi, lo, s
for (M, K, N) in [(32, 16, 32), (32, 16, 96), (32, 16, 64), (32, 32, 96), (128, 96, 96)]:
    x = torch.randn(M, K, device=dev) * 1.5
    w = torch.randn(N, K, device=dev)
    sw, sx = amax_of(w), amax_of(x)
    wp = ops.conv3x3_pack(w.view(N, K, 1, 1), sw, False)
    y = ops.tok_gemm(x, wp, N, sx, sw).double()
    xh, xl, s1 = split(x, x.abs().max().double()); wh, wl, s2 = split(w, w.abs().max().double())
    full = (xh @ wh.t() + xh @ wl.t() + xl @ wh.t()) / (s1 * s2)
    variants = {"exact": x.double() @ w.double().t(), "3 terms": full, "hh": xh @ wh.t() / (s1 * s2),
                "hh+hl": (xh @ wh.t() + xh @ wl.t()) / (s1 * s2), "hh+lh": (xh @ wh.t() + xl @ wh.t()) / (s1 * s2)}
    den = variants["exact"].abs().max()
    print((M, K, N), {k: f"{((y - v).abs().max() / den).item():.1e}" for k, v in variants.items()},
          "per tile:", [f"{((y - full)[:, 32 * t:32 * t + 32].abs().max() / den).item():.1e}" for t in range(N // 32)])
