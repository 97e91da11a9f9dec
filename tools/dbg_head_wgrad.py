"""Weight gradient of the head convolution (12 x 720 x 128 x 256) alone, for PMC passes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd
from mscs_amd.models import ops
dev = torch.device("cuda:0")
n, c, h, w = 12, 720, 128, 256
x = torch.randn(n, c, h, w, device=dev).relu_(); gy = torch.randn(n, c, h, w, device=dev) * 1e-4
for _ in range(3):
    ops.conv3x3_wgrad(x, gy)
torch.cuda.synchronize()
