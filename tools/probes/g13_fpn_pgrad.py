"""Per-parameter gradient error of the G13 UPerNet-FPN fixture against its fp64 record (which tensors carry tests/test_models.py's pgrad)."""
import os, sys, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_models as tm
from conftest import GOLDEN
name, tag = "G13_module_upernet_fpn", "f64_"
dev = torch.device("cuda:0")
z = np.load(os.path.join(GOLDEN, name + ".npz"))
mod = tm._module_under_test(name, dev)
tm.fill_state_dict_(mod)
mod.train().to(dev)
xs = [tm.model_input(tuple(int(v) for v in sh), seed=11 + i).to(dev).requires_grad_(True) for i, sh in enumerate(z["input_shapes"])]
print("inputs", [tuple(x.shape) for x in xs])
outs = tm._flatten(mod(list(xs)) if len(xs) > 1 else mod(xs[0]))
tm._probe_loss(outs).backward()
names = [k for k, _ in mod.named_parameters()]
grads = [p.grad.float().cpu() for _, p in mod.named_parameters()]
amax = z[tag + "pgrad_abs_max"]
step = int(z[tag + "pgrad_step"])
flat = torch.cat([g.flatten() for g in grads]).numpy()
owner = np.repeat(np.arange(len(grads)), [g.numel() for g in grads])
bounds = np.repeat(np.maximum(amax, 1e-3 * float(amax.max())), [g.numel() for g in grads])
err = np.abs(flat[::step] - z[tag + "pgrad_sample"]) / bounds[::step]
own = owner[::step]
for i, n in enumerate(names):
    m = own == i
    if m.any() and err[m].max() > 1e-3:
        print(f"{err[m].max():.2e}  {n:40s} shape {tuple(grads[i].shape)} |g|max {amax[i]:.3e} (block max {amax.max():.3e})")

# conv_last.0.0.weight [256, 1024, 3, 3]: the error by input-channel slice (P2 | P5 | P4 | P3 in the concatenation order)
i = names.index("conv_last.0.0.weight")
start = sum(g.numel() for g in grads[:i])
pos = np.arange(0, flat.size, step)
m = own == i
ci = ((pos[m] - start) // 9) % 1024
for lo_, nm in ((0, "P2 fine"), (256, "P5 8x"), (512, "P4 4x"), (768, "P3 2x")):
    sel = (ci >= lo_) & (ci < lo_ + 256)
    print(f"  conv_last slice {nm}: max err / tensor max {err[m][sel].max():.2e}   max |sample| {np.abs(z[tag + 'pgrad_sample'][m][sel]).max():.2e}")

# ---- every split-f16 GEMM of the backward against float64 on its own operands (which product is off, and by how much)
if os.environ.get("GEMM_CHECK"):
    from mscs_amd.models import ops
    orig = ops.gemm_f16x3
    def checked(a, akm, lda, b, bkm, ldb, M, N, K, out, ldc, a_amax, b_amax, bias=None, batch=1, strides=(0, 0, 0), **kw):
        r = orig(a, akm, lda, b, bkm, ldb, M, N, K, out, ldc, a_amax, b_amax, bias=bias, batch=batch, strides=strides, **kw)
        if kw.get("accumulate") or bias is not None:
            return r
        torch.cuda.synchronize()
        errs = []
        for z in range(batch):
            A = a.flatten()[z * strides[0]:] if strides[0] else a.flatten()
            B = b.flatten()[z * strides[1]:] if strides[1] else b.flatten()
            A2 = (A[:M * lda].view(M, lda)[:, :K] if akm else A[:K * lda].view(K, lda)[:, :M].t()).double()
            B2 = (B[:N * ldb].view(N, ldb)[:, :K] if bkm else B[:K * ldb].view(K, ldb)[:, :N].t()).double()
            want = A2 @ B2.t()
            C = out.flatten()[z * strides[2]:][:M * ldc].view(M, ldc)[:, :N].double()
            lib = (A2.float() @ B2.float().t()).double()
            errs.append(((C - want).abs().max().item(), (lib - want).abs().max().item(), want.abs().max().item(),
                         (A2.abs() @ B2.abs().t()).max().item()))
        e = max(x[0] for x in errs); l = max(x[1] for x in errs); w = max(x[2] for x in errs); s1 = max(x[3] for x in errs)
        print(f"gemm M {M} N {N} K {K} batch {batch} akm {akm} bkm {bkm}: err {e:.2e} (fp32 mm {l:.2e}) max|C| {w:.2e} max sum|terms| {s1:.2e}"
              f"  a_amax {float(a_amax.max()):.3e} true {float(a.abs().max()):.3e}  b_amax {float(b_amax.max()):.3e} true {float(b.abs().max()):.3e}")
        return r
    ops.gemm_f16x3 = checked
    for p in mod.parameters():
        p.grad = None
    xs = [x.detach().clone().requires_grad_(True) for x in xs]
    outs = tm._flatten(mod(list(xs)))
    print("---- backward GEMMs")
    tm._probe_loss(outs).backward()
