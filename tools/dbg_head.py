"""HRNet-W48 head convolution (12 x 720 x 128 x 256, 720 -> 720) on the direct kernels against the f16 GEMM path.
    python tools/dbg_head.py [--fwd-only]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd
from mscs_amd.models import ops
from mscs_amd.models.amax import amax_of
dev = torch.device("cuda:0")
def timeit(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
n, c, h, w = 12, 720, 128, 256
x = torch.randn(n, c, h, w, device=dev).relu_(); wt = torch.randn(c, c, 3, 3, device=dev) * 0.02
gy = torch.randn(n, c, h, w, device=dev) * 1e-4
sx, sw = amax_of(x), amax_of(wt)
wp = ops.conv3x3_pack(wt, sw)
out = torch.empty(n, c, h, w, device=dev)
flops = 2.0 * n * c * c * 9 * h * w
fwd_only = "--fwd-only" in sys.argv
for (r, p) in (((3, 4),) if fwd_only else ((3, 4), (2, 4), (3, 2))):
    t = timeit(lambda: ops.conv3x3_launch(x, wp, c, sx, sw, out, r, p))
    print(f"head fwd direct ({r},{p}): {t:.2f} ms ({flops/t/1e9:.0f} TF)", flush=True)
if fwd_only:
    sys.exit(0)
t = timeit(lambda: ops.conv3x3_wgrad(x, gy), 3)
print(f"head wgrad direct: {t:.2f} ms ({flops/t/1e9:.0f} TF)")
conv = torch.nn.Conv2d(c, c, 3, padding=1).to(dev)
def f16x3():
    xi = x.clone().requires_grad_(True)
    y = ops.conv3x3_f16x3(xi, conv); y.backward(gy)
t = timeit(f16x3, 3)
print(f"head f16x3 GEMM path fwd+bwd: {t:.2f} ms")
