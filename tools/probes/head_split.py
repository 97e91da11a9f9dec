"""Head convolution of HRNet-W48 at the benchmark shape: the split form (ops.conv3x3_over_upsampled) against the direct
convolution of the materialised concatenation, forward + backward, per-kernel times under rocprofv3 (tools/prof_cmd.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd  # noqa: E402,F401
from mscs_amd.models import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
n, H, W = 12, 128, 256
chans = (48, 96, 192, 384)
ts = [torch.randn(n, c, H >> i, W >> i, device=dev).requires_grad_(True) for i, c in enumerate(chans)]
conv = torch.nn.Conv2d(720, 720, 3, 1, 1).to(dev)
ops.use_direct_conv3x3(conv)
gy = torch.randn(n, 720, H, W, device=dev) * 1e-3
mode = sys.argv[1] if len(sys.argv) > 1 else "split"


def run():
    for t in ts:
        t.grad = None
    conv.weight.grad = None
    if mode == "split":
        y = ops.conv3x3_over_upsampled(ts, True, conv.weight, conv.bias)
    else:
        y = conv(ops.upsample_concat(ts, True))
    y.backward(gy)


for _ in range(2):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record()
torch.cuda.synchronize()
print(f"{mode}: {e0.elapsed_time(e1) / 5:.2f} ms per forward + backward")
