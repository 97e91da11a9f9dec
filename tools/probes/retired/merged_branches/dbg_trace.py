"""Which launch is the first to differ from run to run?  (companion of dbg_merged.py)

    python tools/probes/dbg_trace.py [interleave] [runs=N] [group]

Wraps the backward of the fused norm and of the direct convolution: every input and output of every node gets an integer checksum
(sum of its 32-bit words) computed ON THE NODE'S STREAM right behind the node's kernels -- no host synchronisation until the run
is over.  Runs the exchange module N times on the same data and prints, per differing run, the first entries (in execution order)
whose checksum is not the one of run 0."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mscs_amd  # noqa: F401,E402
from test_merged_branches import _module  # noqa: E402
from mscs_amd.models import fused_bn, ops_conv  # noqa: E402
from mscs_amd.models import amax as _amax  # noqa: E402

dev = torch.device("cuda:0")
LOG = []            # (label, checksum tensor)


def cks(t):
    if t is None:
        return None
    t = t.detach()
    if not t.is_contiguous():
        t = t.contiguous()
    v = t.view(-1)
    if v.element_size() == 8:
        return v.view(torch.int64).sum()
    return v.view(torch.int32).sum(dtype=torch.int64)


SEL = os.environ.get("TRACE_SEL", "")            # comma list of substrings: only matching labels are traced ("" = all)
SIDE = os.environ.get("TRACE_SIDE", "0") == "1"     # checksums on a separate stream (tensors kept alive until the run is over)
KEEP = []
_side = torch.cuda.Stream(dev) if SIDE else None


def note(label, t):
    if t is None or (SEL and not any(k in label for k in SEL.split(","))):
        return
    if SIDE:
        cur = torch.cuda.current_stream()
        _side.wait_stream(cur)
        KEEP.append(t)
        with torch.cuda.stream(_side):
            LOG.append((label, cks(t)))
        return
    LOG.append((label, cks(t)))


def tagbuf(t):
    got = getattr(t, "_dcl_amax", None)
    return got[1] if got is not None and got[0] == t._version else None


COUNT = {"bn": 0, "conv": 0}
_bn_bwd = fused_bn._FusedBNFunction.backward
_cv_bwd = ops_conv._Conv3x3Direct.backward


def bn_backward(ctx, dy):
    k = COUNT["bn"]
    COUNT["bn"] += 1
    x, y, weight, bias, mean, invstd = ctx.saved_tensors
    name = f"bn#{k} C={x.shape[1]} HW={x.shape[2]}x{x.shape[3]} res={ctx.has_res} stream={torch.cuda.current_stream().cuda_stream:#x}"
    for lab, t in (("dy", dy), ("dy.amax", tagbuf(dy)), ("x", x), ("y/mask", y), ("gamma", weight), ("beta", bias), ("mean", mean), ("invstd", invstd)):
        note(f"{name} in:{lab}", t)
    out = _bn_bwd(ctx, dy)
    dx, dres, dgamma, dbeta = out[0], out[1], out[2], out[3]
    if ctx.token is not None:
        dres = ctx.token.dres
    for lab, t in (("dx", dx), ("dx.amax", tagbuf(dx)), ("dres", dres), ("dgamma", dgamma), ("dbeta", dbeta),
                   ("dy-after", dy), ("x-after", x)):
        note(f"{name} out:{lab}", t)
    return out


def conv_backward(ctx, gy):
    k = COUNT["conv"]
    COUNT["conv"] += 1
    x, weight = ctx.saved_tensors
    name = f"conv#{k} {weight.shape[1]}->{weight.shape[0]} HW={x.shape[2]}x{x.shape[3]} stream={torch.cuda.current_stream().cuda_stream:#x}"
    wamax, wp, wpt = ctx.mod.packed_weights()
    tok = ctx.token.dres if ctx.token is not None else None
    for lab, t in (("gy", gy), ("gy.amax", tagbuf(gy)), ("x", x), ("x.amax", tagbuf(x)), ("w", weight), ("wamax", wamax), ("wpt", wpt), ("addend", tok)):
        note(f"{name} in:{lab}", t)
    out = _cv_bwd(ctx, gy)
    for lab, t in (("gx", out[0]), ("gw", out[1]), ("gy-after", gy)):
        note(f"{name} out:{lab}", t)
    return out


fused_bn._FusedBNFunction.backward = staticmethod(bn_backward)
ops_conv._Conv3x3Direct.backward = staticmethod(conv_backward)

hm, mod, ch = _module(4, dev)
state = {k: v.clone() for k, v in mod.state_dict().items()}
hw = tuple(int(v) for v in os.environ.get("DBG_HW", "64,96").split(","))
xs0 = [torch.randn(2, c, hw[0] >> i, hw[1] >> i, device=dev) for i, c in enumerate(ch)]
hm._MERGE_INTERLEAVE = "interleave" in sys.argv
nrun = next((int(a[5:]) for a in sys.argv if a.startswith("runs=")), 16)
merged = "group" not in sys.argv
runs = []
for j in range(nrun):
    mod.load_state_dict(state)
    mod.zero_grad(set_to_none=True)
    xs = [x.clone().requires_grad_(True) for x in xs0]
    hm._MERGE_BRANCHES = merged
    fused_bn.FORCE_GROUP = not merged
    LOG.clear()
    KEEP.clear()
    COUNT["bn"] = COUNT["conv"] = 0
    try:
        outs = mod(list(xs))
    finally:
        fused_bn.FORCE_GROUP = False
    sum((o * torch.cos(torch.arange(o.numel(), device=dev).view(o.shape) * 0.37)).mean() for o in outs).backward()
    torch.cuda.synchronize()
    labels = [l for l, _ in LOG]
    vals = torch.stack([c for _, c in LOG]).cpu().tolist()
    final = {n: cks(p.grad).item() for n, p in mod.named_parameters() if p.grad is not None}
    final.update({f"xgrad{i}": cks(x.grad).item() for i, x in enumerate(xs)})
    runs.append((labels, vals, final))
ndiff = 0
for j in range(1, nrun):
    l0, v0, f0 = runs[0]
    lj, vj, fj = runs[j]
    assert l0 == lj, "different node sequence"
    bad = [i for i in range(len(v0)) if v0[i] != vj[i]]
    fbad = [n for n in f0 if f0[n] != fj[n]]
    if bad or fbad:
        ndiff += 1
        print(f"run {j}: {len(bad)} of {len(v0)} trace entries differ, {len(fbad)} final tensors differ; first entries:")
        for i in bad[:6]:
            print("     ", i, l0[i])
print(f"SUMMARY {sys.argv[1:]}: {ndiff} of {nrun - 1} runs differ ({len(runs[0][0])} trace entries per run)")
