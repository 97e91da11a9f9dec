import os, sys, importlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd
hm = importlib.import_module("mscs_amd.models.HRNet")
graph = {"backbone": "hrnet48", "pretrained": False, "dataset": "CITYSCAPES", "align_corners": True,
         "ms_projector": {"mlp": [[1, -1, 1]], "scales": 3, "d": 256, "use_bn": True, "before_context": True}}
torch.manual_seed(5)
dev = torch.device("cuda:0")
model = hm.HRNet(graph, 1).to(dev).train()
shape = tuple(int(v) for v in sys.argv[2].split("x")) if len(sys.argv) > 2 else (2, 3, 128, 256)
x = torch.randn(*shape, device=dev)
hm._BRANCH_STREAMS = sys.argv[1] == "1"
for it in range(3):
    model.zero_grad(set_to_none=True)
    out, proj = model(x)
    (out.square().mean() + sum(p.square().mean() for p in proj)).backward()
    torch.cuda.synchronize()
    print("iter", it, "ok", float(out.abs().mean()), flush=True)
