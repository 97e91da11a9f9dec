"""hrnet18 / hrnet32 (channel counts that are not multiples of 16 or 32) through the direct kernels against the
library path: logits and gradients of one training step.  python tools/check_small_backbones.py  (GPU)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd
from mscs_amd.models import HRNet
dev = torch.device("cuda:0")
for bb in ("hrnet18", "hrnet32"):
    graph = {"backbone": bb, "pretrained": False, "dataset": "CITYSCAPES", "align_corners": True,
             "ms_projector": {"mlp": [[1, -1, 1]], "scales": 3, "d": 256, "use_bn": True, "before_context": True}}
    torch.manual_seed(0)
    m = HRNet(graph, 1).to(dev).train()
    ref = HRNet(dict(graph, branch_conv="library", head_conv="library", fused_bn=False), 1).to(dev).train()
    ref.load_state_dict(m.state_dict())
    x = torch.randn(2, 3, 96, 160, device=dev)
    for mod in (m, ref):
        mod.zero_grad(set_to_none=True)
    o, p = m(x); (o.square().mean() + sum(q.square().mean() for q in p)).backward()
    o2, p2 = ref(x); (o2.square().mean() + sum(q.square().mean() for q in p2)).backward()
    torch.cuda.synchronize()
    err = ((o - o2).abs().max() / o2.abs().max()).item()
    gerr = max(((a.grad - b.grad).abs().max() / (b.grad.abs().max() + 1e-20)).item() for a, b in zip(m.parameters(), ref.parameters()) if a.grad is not None and b.grad is not None and b.grad.abs().max() > 1e-6)
    print(bb, "out rel err", err, "max grad rel err", gerr, "finite", all(torch.isfinite(a.grad).all().item() for a in m.parameters() if a.grad is not None))
