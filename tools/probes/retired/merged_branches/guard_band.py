"""Out-of-bounds WRITE detector for the library's kernels (no GPU sanitizer on this pool).

    DBG_HW=128,256 python tools/probes/guard_band.py [merged]

torch.empty / torch.empty_like are replaced (Python level: every output / workspace the package's autograd functions allocate)
by versions that put a 64-KiB guard zone filled with a byte pattern on either side of the tensor.  One exchange module (W48
channels, all fuse rows: stride-2 chains, 1x1 + up-sampling) runs forward + backward; afterwards every guard byte must still
hold the pattern.  A kernel that writes before the first or past the last element of one of its buffers shows up here with
the shape of the tensor it damaged."""
import math
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mscs_amd  # noqa: F401,E402
from test_merged_branches import _module  # noqa: E402

G = 65536
PAT = 0xA5
REG = []
real_empty, real_empty_like = torch.empty, torch.empty_like


def g_empty(*size, dtype=None, device=None, **kw):
    shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
    dtype = dtype or torch.float32
    if device is None or torch.device(device).type != "cuda" or kw.get("pin_memory") or kw.get("memory_format") not in (None, torch.contiguous_format):
        return real_empty(*size, dtype=dtype, device=device, **kw)
    es = real_empty(0, dtype=dtype).element_size()
    nbytes = int(math.prod(shape)) * es
    pad = (-nbytes) % 256
    raw = real_empty(nbytes + pad + 2 * G, dtype=torch.uint8, device=device)
    raw.fill_(PAT)
    REG.append((raw, nbytes, shape, dtype, "".join(traceback.format_stack(limit=4)[:-1])))
    return raw[G:G + nbytes].view(dtype).view(shape)


def g_empty_like(t, **kw):
    if not t.is_cuda or not t.is_contiguous() or kw.get("memory_format") not in (None, torch.contiguous_format, torch.preserve_format):
        return real_empty_like(t, **kw)
    return g_empty(tuple(t.shape), dtype=kw.get("dtype", t.dtype), device=t.device)


dev = torch.device("cuda:0")
hm, mod, ch = _module(4, dev)
hm._MERGE_BRANCHES = "merged" in sys.argv
hw = tuple(int(v) for v in os.environ.get("DBG_HW", "128,256").split(","))
n = int(os.environ.get("DBG_N", "2"))
xs = [torch.randn(n, c, hw[0] >> i, hw[1] >> i, device=dev).requires_grad_(True) for i, c in enumerate(ch)]
torch.empty, torch.empty_like = g_empty, g_empty_like
try:
    outs = mod(list(xs))
    sum((o * torch.cos(torch.arange(o.numel(), device=dev).view(o.shape) * 0.37)).mean() for o in outs).backward()
    torch.cuda.synchronize()
finally:
    torch.empty, torch.empty_like = real_empty, real_empty_like
bad = 0
for raw, nbytes, shape, dtype, where in REG:
    lo, hi = raw[:G], raw[G + nbytes + ((-nbytes) % 256):]
    for name, z in (("before", lo), ("after", hi)):
        d = (z != PAT).nonzero().flatten()
        if d.numel():
            bad += 1
            print(f"GUARD DAMAGED {name} tensor {shape} {dtype}: {d.numel()} bytes, offsets {d[0].item()}..{d[-1].item()} of the {name} zone\n{where}")
print(f"{len(REG)} guarded allocations, {bad} damaged guard zones (merged={hm._MERGE_BRANCHES}, {n}x{hw})")
