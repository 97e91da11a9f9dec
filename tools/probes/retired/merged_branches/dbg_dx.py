"""Structure of the run-to-run difference in the norm backward's dx (companion of dbg_trace.py).

    python tools/probes/dbg_dx.py [runs=N]

Keeps a clone of dx of every residual norm's backward of branch 0 (taken on the node's stream) for N runs of the exchange module in
the interleaved order; for the first node whose dx differs from run 0 it prints where the difference sits (channels, images, rows)
and what it looks like (constant per plane: the mean-of-gradient term; proportional to x - mean: the projection term)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mscs_amd  # noqa: F401,E402
from test_merged_branches import _module  # noqa: E402
from mscs_amd.models import fused_bn  # noqa: E402

dev = torch.device("cuda:0")
DUMP = []
INPUTS = {}
COUNT = [0]
_bn_bwd = fused_bn._FusedBNFunction.backward
RUN = [0]


def bn_backward(ctx, dy):
    k = COUNT[0]
    COUNT[0] += 1
    out = _bn_bwd(ctx, dy)
    if ctx.has_res and torch.cuda.current_stream().cuda_stream == 0:
        DUMP.append((k, out[0].clone()))
        if RUN[0] == 0:
            x, y, weight, bias, mean, invstd = ctx.saved_tensors
            INPUTS[k] = (dy.clone(), x.clone(), mean.clone(), invstd.clone(), weight.clone(), out[2].clone(), out[3].clone())
    return out


fused_bn._FusedBNFunction.backward = staticmethod(bn_backward)
hm, mod, ch = _module(4, dev)
state = {k: v.clone() for k, v in mod.state_dict().items()}
hw = tuple(int(v) for v in os.environ.get("DBG_HW", "64,96").split(","))
xs0 = [torch.randn(2, c, hw[0] >> i, hw[1] >> i, device=dev) for i, c in enumerate(ch)]
hm._MERGE_INTERLEAVE = True
hm._MERGE_BRANCHES = True
nrun = next((int(a[5:]) for a in sys.argv if a.startswith("runs=")), 32)
runs = []
for j in range(nrun):
    RUN[0] = j
    mod.load_state_dict(state)
    mod.zero_grad(set_to_none=True)
    xs = [x.clone().requires_grad_(True) for x in xs0]
    DUMP.clear()
    COUNT[0] = 0
    outs = mod(list(xs))
    sum((o * torch.cos(torch.arange(o.numel(), device=dev).view(o.shape) * 0.37)).mean() for o in outs).backward()
    torch.cuda.synchronize()
    runs.append(list(DUMP))
shown = 0
nbad = 0
for j in range(1, nrun):
    first = next(((k, a, b) for (k, a), (_, b) in zip(runs[0], runs[j]) if not torch.equal(a, b)), None)
    if first is None:
        continue
    nbad += 1
    if shown >= 4:
        continue
    shown += 1
    k, a, b = first
    d = (b - a).double()
    dy, x, mean, invstd, gamma, dgamma, dbeta = INPUTS[k]
    N, C, H, W = a.shape
    chans = [c for c in range(C) if d[:, c].abs().max() > 0]
    print(f"run {j}: first differing dx at node bn#{k}, shape {tuple(a.shape)}, max|dx| {a.abs().max().item():.3e}; channels with a difference: {chans}")
    for c in chans[:4]:
        dc = d[:, c]
        nz = dc != 0
        rows = nz.any(dim=2)
        per_n = [(n, int(rows[n].nonzero().min()), int(rows[n].nonzero().max()), int(nz[n].sum())) for n in range(N) if rows[n].any()]
        xc = (x[:, c].double() - mean[c].double()) * invstd[c].double()
        sel = dc[nz]
        # least squares  d = alpha + beta * xhat  over the differing elements
        A = torch.stack([torch.ones_like(xc[nz]), xc[nz]], 1)
        sol = torch.linalg.lstsq(A, sel.unsqueeze(1)).solution.squeeze(1)
        resid = (A @ sol - sel).abs().max().item()
        kf = (invstd[c] * gamma[c]).item()
        print(f"   channel {c}: {int(nz.sum())} of {dc.numel()} elements differ; (image, first row, last row, count) {per_n}; "
              f"diff min {sel.min().item():.3e} max {sel.max().item():.3e}; fit d = a + b xhat: a {sol[0].item():.3e} b {sol[1].item():.3e} "
              f"(residual {resid:.1e}); k = invstd gamma = {kf:.3e}; -a/k = {(-sol[0] / kf).item():.3e} -b/k = {(-sol[1] / kf).item():.3e}; "
              f"this channel's sum g / count = {(dbeta[c] / (N * H * W)).item():.3e}, sum g xhat / count = {(dgamma[c] / (N * H * W)).item():.3e}")
        flat = d.view(-1).nonzero().view(-1)
        hw4 = H * W // 4
        for n in range(N):
            idx = nz[n].view(-1).nonzero().view(-1)
            if idx.numel() == 0:
                continue
            p0, p1 = int(idx.min()), int(idx.max())
            jv = n * hw4 + p0 // 4
            bx, r = jv // 1024, jv % 1024
            addr = a.data_ptr() + (((n * C + c) * H * W) + p0) * 4
            print(f"      image {n}: plane elements {p0}..{p1} ({idx.numel()} differ, contiguous {p1 - p0 + 1 == idx.numel()}); vector j {jv}: chunk {bx}, u {r // 256}, "
                  f"thread {r % 256} (wave {(r % 256) // 64} lane {r % 64}); byte address {addr:#x} (mod 64: {addr % 64}, mod 128: {addr % 128}, mod 4096: {addr % 4096}, "
                  f"mod 2^20: {addr % (1 << 20):#x})")
print(f"SUMMARY: {nbad} of {nrun - 1} runs differ")
