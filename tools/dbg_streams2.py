import os, sys, importlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd
hm = importlib.import_module("mscs_amd.models.HRNet")
graph = {"backbone": "hrnet48", "pretrained": False, "dataset": "CITYSCAPES", "align_corners": True,
         "ms_projector": {"mlp": [[1, -1, 1]], "scales": 3, "d": 256, "use_bn": True, "before_context": True}}
torch.manual_seed(5)
dev = torch.device("cuda:0")
model = hm.HRNet(graph, 1).to(dev).train()
x = torch.randn(2, 3, 128, 256, device=dev)
def run(flag):
    hm._BRANCH_STREAMS = flag
    model.zero_grad(set_to_none=True)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    out, proj = model(x)
    (out.square().mean() + sum(p.square().mean() for p in proj)).backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    model.load_state_dict(state)
    return out.detach().clone(), grads
def dist(a, b): return (a - b).abs().max().item() / (b.abs().max().item() + 1e-20)
o0, g0 = run(False); o0b, g0b = run(False); o0c, g0c = run(False); o1, g1 = run(True); o2, g2 = run(True)
print("out noise", dist(o0b, o0), dist(o0c, o0), "stream", dist(o1, o0), dist(o2, o0))
rows = sorted(((dist(g1[n], g0[n]), dist(g2[n], g0[n]), dist(g0b[n], g0[n]), dist(g0c[n], g0[n]), n) for n in g0), reverse=True)
for r in rows[:12]: print("%.2e %.2e | noise %.2e %.2e %s" % r)
rows2 = sorted(((dist(g0b[n], g0[n]), n) for n in g0), reverse=True)
print("largest single-stream noise:", ["%.2e %s" % r for r in rows2[:5]])
