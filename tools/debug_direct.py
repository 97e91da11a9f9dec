import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd
from mscs_amd.models import HRNet, ops
graph = {"backbone": "hrnet48", "pretrained": False, "dataset": "CITYSCAPES", "align_corners": True,
         "ms_projector": {"mlp": [[1, -1, 1]], "scales": 3, "d": 256, "use_bn": True, "before_context": True}}
m = HRNet(graph, 1).cuda().train()
n = sum(isinstance(x, ops.DirectConv2d) for x in m.modules())
print("direct modules", n)
cnt = {"e": 0, "f": 0}
orig = ops.DirectConv2d.eligible
def elig(self, x):
    r = orig(self, x)
    cnt["e" if r else "f"] += 1
    if not r and cnt["f"] < 4:
        print("ineligible", self, x.shape, x.dtype, x.is_contiguous(), torch.is_autocast_enabled())
    return r
ops.DirectConv2d.eligible = elig
x = torch.randn(2, 3, 256, 512, device="cuda")
out = m(x)
print(cnt)
(out[0].mean() + sum(p.mean() for p in out[1])).backward()
print("ok")
