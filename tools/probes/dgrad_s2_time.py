"""Data gradient of the stride-2 convolutions (k_conv3x3_pm) on the step's shapes: HIP-event time per launch and distance to float64."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd  # noqa: E402,F401
from mscs_amd.models import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
print("library:", os.environ.get("DCL_LIB_PATH", "product"))
for (n, ci, co, h, w) in [(12, 48, 96, 128, 256), (12, 48, 48, 128, 256), (12, 64, 64, 256, 512), (12, 96, 192, 64, 128), (2, 32, 48, 18, 40)]:
    wt = torch.randn(co, ci, 3, 3, device=dev) * 0.05
    gy = torch.randn(n, co, (h - 1) // 2 + 1, (w - 1) // 2 + 1, device=dev)
    for _ in range(3):
        gx = ops.conv3x3_direct(gy, wt, transposed=True, stride=2, out_hw=(h, w))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        gx = ops.conv3x3_direct(gy, wt, transposed=True, stride=2, out_hw=(h, w))
    e1.record()
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_input((n, ci, h, w), wt.double(), gy.double(), stride=2, padding=1)
    err = ((gx.double() - ref).abs().max() / ref.abs().max()).item()
    print(f"  {n}x({ci}<-{co})x{h}x{w}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us   distance to float64 {err:.1e}   checksum {gx.double().sum().item():.10e}")
