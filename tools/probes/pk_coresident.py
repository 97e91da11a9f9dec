"""Does the norm backward (packed-FP32 code) return different bits when ANOTHER kernel shares its compute units?

    [DCL_LIB_PATH=variant.so] python tools/probes/pk_coresident.py [iters=N]

Stream A: the backward of one FusedBatchNorm2d (+ residual + ReLU, 2 x 48 x 64 x 96: the shape and form of the launches that differ
in tools/probes/dbg_dx.py) again and again on the same inputs, every dx compared bitwise with the one computed alone.  Stream B,
concurrently: nothing | the f16x3 convolution (matrix pipe) | norm statistics (vector ALU, no matrix instructions) | a plain
ATen element-wise kernel."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mscs_amd  # noqa: F401,E402
from mscs_amd.models import fused_bn, ops  # noqa: E402
from mscs_amd.models.amax import amax_of  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
NB = next((int(a[3:]) for a in sys.argv if a.startswith("nb=")), 8)
iters = next((int(a[6:]) for a in sys.argv if a.startswith("iters=")), 60)
bn = fused_bn.FusedBatchNorm2d(48).to(dev).train()
with torch.no_grad():
    bn.weight.uniform_(0.5, 1.5)
    bn.bias.uniform_(-0.5, 0.5)
x = torch.randn(2, 48, 64, 96, device=dev, requires_grad=True)
res = torch.randn(2, 48, 64, 96, device=dev)
dy = torch.randn(2, 48, 64, 96, device=dev) * 1e-4
y = bn(x, residual=res, relu=True)
ref = torch.autograd.grad(y, x, dy, retain_graph=True)[0].clone()
torch.cuda.synchronize()

# stream B's work
cx = torch.randn(12, 96, 64, 128, device=dev).relu_()
cw = torch.randn(96, 96, 3, 3, device=dev) * 0.03
sx, sw = amax_of(cx), amax_of(cw)
wp = ops.conv3x3_pack(cw, sw)
cout = torch.empty_like(cx)
bx = torch.randn(12, 96, 64, 128, device=dev)
L = mscs_amd._lib.lib() if hasattr(mscs_amd, "_lib") else None
from mscs_amd import _lib  # noqa: E402
L = _lib.lib()
ns = L.dcl_bn_num_slices(12, 96)
part = torch.empty(96 * ns * 2, device=dev)
rm = torch.zeros(96, device=dev)
piv = torch.empty(96, device=dev)


def b_conv():
    ops.conv3x3_launch(cx, wp, 96, sx, sw, cout)


def b_stats():
    _lib.check(L.dcl_bn_stats_part(_lib.ptr(bx), 12, 96, 64 * 128, _lib.ptr(part), _lib.ptr(rm), _lib.ptr(piv), _lib.stream_ptr(dev)), "stats")


def b_aten():
    torch.mul(bx, 1.0001, out=cout)


gy96 = torch.randn(12, 96, 64, 128, device=dev)
gy192 = torch.randn(12, 192, 32, 64, device=dev)
x192 = torch.randn(12, 192, 32, 64, device=dev).relu_()


def b_wgrad96():
    ops.conv3x3_wgrad(cx, gy96)


def b_wgrad192():
    ops.conv3x3_wgrad(x192, gy192)


# (round 5 also ran the merged coarse-block backward of models/merged.py as a neighbour; that schedule is retired:
# tools/probes/retired/merged_branches/)
sb = torch.cuda.Stream(dev)
sa = torch.cuda.Stream(dev)
# victim: the norm backward of this library (default) or ATen element-wise kernels (`aten`: whatever PyTorch's own build made of
# them -- not this library's code, not built with NOPK)
va, vb, vc = (torch.randn(2, 48, 64, 96, device=dev) for _ in range(3))
if "aten" in sys.argv:
    def victim():
        return torch.addcmul(va, vb, vc, value=0.37).mul_(1.7).add_(vb, alpha=-0.21)
elif "bnfwd" in sys.argv:                       # the norm's forward (statistics + apply with residual and ReLU)
    def victim():
        with torch.no_grad():
            return bn(x, residual=res, relu=True)
elif "bnplain" in sys.argv:                     # backward of a norm + ReLU without residual (mask recomputed from x)
    y2 = bn(x, relu=True)

    def victim():
        return torch.autograd.grad(y2, x, dy, retain_graph=True)[0]
elif "upsample" in sys.argv:                    # bilinear up-sampling + add (csrc/dcl_resize.hip)
    ub = torch.randn(2, 48, 128, 192, device=dev)

    def victim():
        return ops.upsample_bilinear(va, (128, 192), True) + 0
else:
    def victim():
        return torch.autograd.grad(y, x, dy, retain_graph=True)[0]
ref = victim().clone()
torch.cuda.synchronize()
CASES = (("nothing", None), ("f16x3 convolution (MFMA)", b_conv), ("weight gradient 96 ch (LDS-DMA + MFMA)", b_wgrad96),
         ("weight gradient 192 ch", b_wgrad192),          ("norm statistics (VALU)", b_stats), ("ATen mul", b_aten))
c48 = torch.randn(12, 48, 128, 256, device=dev).relu_()
w48 = torch.randn(48, 48, 3, 3, device=dev) * 0.05
s48x, s48w = amax_of(c48), amax_of(w48)
wp48 = ops.conv3x3_pack(w48, s48w)
o48 = torch.empty_like(c48)
g48 = torch.randn(12, 48, 128, 256, device=dev)
lx = torch.randn(16384, 384, device=dev)
lw = torch.randn(1536, 384, device=dev) * 0.05


def b_conv48():
    ops.conv3x3_launch(c48, wp48, 48, s48x, s48w, o48)


def b_wgrad48():
    ops.conv3x3_wgrad(c48, g48)


def b_gemm():
    ops.linear_f16x3(lx, lw)


if "mfma" in sys.argv:
    CASES = (("convolution 48 ch, two workgroups per CU (32x32x16)", b_conv48), ("GEMM 16384 x 384 -> 1536 (32x32x16)", b_gemm),
             ("weight gradient 48 ch (16x16x32)", b_wgrad48), ("weight gradient 96 ch (16x16x32)", b_wgrad96))
if "wgrad" in sys.argv:
    CASES = (("weight gradient 96 ch (LDS-DMA + MFMA)", b_wgrad96), ("weight gradient 192 ch", b_wgrad192))
print("library:", os.environ.get("DCL_LIB_PATH", "product"))
for name, fb in CASES:
    bad = total = 0
    worst = 0.0
    for it in range(iters):
        outs = []
        if fb is not None:
            with torch.cuda.stream(sb):
                for _ in range(NB):
                    fb()
        with torch.cuda.stream(sa):
            for _ in range(12):
                outs.append(victim())
        torch.cuda.synchronize()
        for o in outs:
            total += 1
            if not torch.equal(o, ref):
                bad += 1
                worst = max(worst, ((o - ref).abs().max() / ref.abs().max()).item())
    print(f"stream B = {name}: {bad} of {total} results on stream A differ from the one computed alone (worst {worst:.1e} of max)", flush=True)
