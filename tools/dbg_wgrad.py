import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd
from mscs_amd.models import ops
dev = torch.device("cuda:0")
for (n, c, h, w) in [(12, 48, 128, 256), (12, 96, 64, 128), (12, 192, 32, 64), (12, 384, 16, 32)]:
    x = torch.randn(n, c, h, w, device=dev); gy = torch.randn_like(x)
    for _ in range(5):
        ops.conv3x3_wgrad(x, gy)
    torch.cuda.synchronize()
