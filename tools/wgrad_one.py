import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import mscs_amd
from mscs_amd.models import ops
dev = torch.device("cuda:0")
c, h, w = [int(v) for v in sys.argv[1].split("x")]
x = torch.randn(12, c, h, w, device=dev).relu_(); gy = torch.randn(12, c, h, w, device=dev) * 1e-3
for _ in range(20):
    ops.conv3x3_wgrad(x, gy)
torch.cuda.synchronize()
