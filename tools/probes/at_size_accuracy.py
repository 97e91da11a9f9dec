"""Max-norm error against float64 AT THE BENCHMARK SIZES for the kernels that split f32 operands in registers with inline asm
right in front of MFMAs (a missing wait state shows up as ~1e-4 errors in one of ~10^5 results: invisible in small tests)."""
import os
import sys

import torch
import torch.nn.functional as F

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import __graft_entry__  # noqa: F401,E402
from mscs_amd.models import ops  # noqa: E402

dev = "cuda"
torch.manual_seed(0)


def rel(a, r):
    return ((a.double() - r).abs().max() / r.abs().max()).item()


for (n, ci, co, h, w, st) in [(12, 48, 48, 128, 256, 1), (12, 96, 96, 64, 128, 1), (12, 192, 192, 32, 64, 1), (12, 384, 384, 16, 32, 1),
                             (4, 144, 720, 128, 256, 1), (12, 64, 64, 128, 256, 1), (12, 48, 96, 128, 256, 2), (12, 64, 64, 256, 512, 2)]:
    x = torch.randn(n, ci, h, w, device=dev).relu_()
    wt = torch.randn(co, ci, 3, 3, device=dev) * 0.1
    ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
    gy = torch.randn(n, co, ho, wo, device=dev) * 1e-3
    conv = torch.nn.Conv2d(ci, co, 3, st, 1, bias=False).to(dev)
    conv.weight.data.copy_(wt)
    ops.use_direct_conv3x3(conv)
    xi = x.clone().requires_grad_(True)
    y = conv(xi)
    y.backward(gy)
    # float64 in slices of the batch to bound memory
    errs = [0.0, 0.0]
    gw64 = torch.zeros(co, ci, 3, 3, dtype=torch.float64, device=dev)
    ymax = gxmax = 0.0
    ey = egx = 0.0
    for b in range(n):
        x64 = x[b:b + 1].double().requires_grad_(True)
        w64 = wt.double().requires_grad_(True)
        y64 = F.conv2d(x64, w64, None, st, 1)
        y64.backward(gy[b:b + 1].double())
        ey = max(ey, (y[b:b + 1].double() - y64).abs().max().item())
        ymax = max(ymax, y64.abs().max().item())
        egx = max(egx, (xi.grad[b:b + 1].double() - x64.grad).abs().max().item())
        gxmax = max(gxmax, x64.grad.abs().max().item())
        gw64 += w64.grad
    print(f"conv {n}x{ci}->{co} {h}x{w} s{st}: y {ey / ymax:.2e} dx {egx / gxmax:.2e} dw {rel(conv.weight.grad, gw64):.2e}", flush=True)
    del x, gy, xi, y
    torch.cuda.empty_cache()

# ---- the contrastive loss at the benchmark size (N = 9 804 anchors per scale): split-f16 sweeps against the f32-MFMA
# sweeps on the same plan (same generator state): loss and every feature gradient
from mscs_amd.losses import DenseContrastiveLossV2_ms  # noqa: E402
from mscs_amd.utils import set_verbosity  # noqa: E402

set_verbosity(40)
gen = torch.Generator().manual_seed(5)
n, H, W, S = 12, 512, 1024, 3
label = torch.randint(0, 20, (n, H, W), generator=gen).to(dev)
feats0 = [torch.randn(n, 256, H // (4 << s), W // (4 << s), generator=gen).to(dev) for s in range(S)]
res = {}
for mode in ("f32", "f16x3"):
    cfg = {"mfma_mode": mode, "dataset": "CITYSCAPES", "experiment": 1, "temperature": 0.1, "scales": S, "weights": [1.0, 0.7, 0.4],
           "cross_scale_contrast": True, "min_views_per_class": 5, "max_views_per_class": 2500, "max_features_total": 10000,
           "label_scaling_mode": "nn"}
    mod = DenseContrastiveLossV2_ms(cfg)
    feats = [f.clone().requires_grad_(True) for f in feats0]
    torch.manual_seed(0)
    loss = mod(label, feats)
    loss.backward()
    res[mode] = (loss.item(), [f.grad.clone() for f in feats])
print(f"loss f32 {res['f32'][0]:.7f} f16x3 {res['f16x3'][0]:.7f} rel {abs(res['f32'][0] - res['f16x3'][0]) / abs(res['f32'][0]):.2e}")
for s in range(S):
    a, b = res["f32"][1][s], res["f16x3"][1][s]
    print(f"  scale {s}: max |dgrad| / max |grad| {((a - b).abs().max() / a.abs().max()).item():.2e}   "
          f"nonzero rows equal: {bool(((a != 0) == (b != 0)).all())}")
