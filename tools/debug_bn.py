import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mscs_amd
from mscs_amd.models.fused_bn import FusedBatchNorm2d
dev = torch.device("cuda:0")
for relu, use_res, shape in [(True, True, (3, 48, 33, 47)), (True, True, (3, 48, 32, 48)), (False, False, (3, 48, 33, 47)), (True, False, (3, 48, 33, 47)), (False, True, (3, 48, 33, 47))]:
    torch.manual_seed(3)
    C = shape[1]
    ref = torch.nn.BatchNorm2d(C, momentum=0.1).to(dev)
    fus = FusedBatchNorm2d(C, momentum=0.1).to(dev)
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5); ref.bias.uniform_(-0.5, 0.5)
    fus.load_state_dict(ref.state_dict())
    x = (torch.randn(shape, device=dev) * 2 + 0.7)
    r = torch.randn(shape, device=dev) if use_res else None
    gy = torch.randn(shape, device=dev)
    outs = []
    for m in (ref, fus):
        xi = x.clone().requires_grad_(True)
        ri = r.clone().requires_grad_(True) if use_res else None
        if m is fus:
            y = m(xi, residual=ri, relu=relu)
        else:
            y = m(xi)
            if use_res: y = y + ri
            if relu: y = torch.relu(y)
        y.backward(gy)
        outs.append(dict(y=y.detach(), dx=xi.grad, dres=ri.grad if use_res else None, dw=m.weight.grad, db=m.bias.grad, rm=m.running_mean.clone(), rv=m.running_var.clone()))
    print(relu, use_res, shape, {k: (float((outs[0][k]-outs[1][k]).abs().max()) if outs[0][k] is not None else None) for k in outs[0]})
print("---- against CPU float64")
for shape in [(3, 48, 33, 47), (3, 48, 32, 48)]:
    torch.manual_seed(3)
    C = shape[1]
    x = (torch.randn(shape, device=dev) * 2 + 0.7); gy = torch.randn(shape, device=dev)
    w = torch.rand(C, device=dev) + 0.5; b = torch.rand(C, device=dev)
    res = {}
    for name in ("cpu64", "torch_gpu", "fused"):
        if name == "cpu64":
            m = torch.nn.BatchNorm2d(C).double(); xi = x.cpu().double().requires_grad_(True); g = gy.cpu().double()
        elif name == "torch_gpu":
            m = torch.nn.BatchNorm2d(C).to(dev); xi = x.clone().requires_grad_(True); g = gy
        else:
            m = FusedBatchNorm2d(C).to(dev); xi = x.clone().requires_grad_(True); g = gy
        with torch.no_grad():
            m.weight.copy_(w); m.bias.copy_(b)
        y = m(xi); y.backward(g)
        res[name] = (xi.grad.double().cpu(), m.weight.grad.double().cpu())
    for name in ("torch_gpu", "fused"):
        print(shape, name, "dx err", float((res[name][0]-res["cpu64"][0]).abs().max()), "dw err", float((res[name][1]-res["cpu64"][1]).abs().max()))
