"""Which lines of this repository call the framework's element-wise / copy operators on large tensors in a training step?

    python tools/aten_sources.py --config 5 [--min-mb 8] [--out gpurun_out/aten_sources_config5.txt]

A TorchDispatchMode logs every aten operator whose largest tensor argument has at least --min-mb megabytes, with shapes, strides
and the innermost repository frames of the Python stack (operators run by autograd's built-in nodes have no repository frame:
they are listed under the node that ran them, "<autograd>").  The library's own kernels are ctypes calls and never appear."""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402


class Log(TorchDispatchMode):
    def __init__(self, min_bytes):
        super().__init__()
        self.min_bytes = min_bytes
        self.rows = collections.OrderedDict()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(s in name for s in ("view", "reshape", "as_strided", "detach", "alias", "permute", "transpose", "expand", "slice",
                                   "select", "unsqueeze", "squeeze", "t.default", "split", "unbind", "empty", "_unsafe_view")):
            return out
        big = 0
        desc = []
        for a in list(args) + ([out] if isinstance(out, torch.Tensor) else []):
            if isinstance(a, torch.Tensor):
                big = max(big, a.numel() * a.element_size())
                desc.append(f"{tuple(a.shape)}{'' if a.is_contiguous() else '/' + str(tuple(a.stride()))}")
        if big < self.min_bytes:
            return out
        frames = [f for f in traceback.extract_stack() if ROOT in f.filename and "aten_sources" not in f.filename]
        where = " <- ".join(f"{os.path.relpath(f.filename, ROOT).replace('eccv2022-multi-scale-and-cross-scale-contrastive-segmentation_amd', 'PKG')}:{f.lineno}"
                            for f in reversed(frames[-3:])) or "<autograd>"
        key = (name, " ".join(desc[:4]), where)
        r = self.rows.setdefault(key, [0, 0])
        r[0] += 1
        r[1] += big
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=5)
    ap.add_argument("--min-mb", type=float, default=8.0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import bench
    sys.argv = ["bench.py", "--config", str(a.config), "--no-cpu-baseline", "--no-eager-step"]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    import mscs_amd  # noqa: F401
    from mscs_amd.managers import HRNetManager, OCRNetManager
    from mscs_amd.utils import set_verbosity
    set_verbosity(40)
    mgr = (OCRNetManager if args.config in (4, 5) else HRNetManager)(bench.step_config(args, 1), autostart=False)
    mgr.setup()
    mgr.model.train()
    gen = torch.Generator().manual_seed(0)
    img = torch.randn(args.batch, 3, args.height, args.width, generator=gen).to(dev)
    lbl = bench.synth_labels(args, args.batch, args.height, args.width, gen).to(dev)
    torch.cuda.synchronize()
    ready = torch.cuda.Event()
    ready.record()

    def step():
        mgr.optimiser.zero_grad(set_to_none=True)
        ret = mgr.forward_step(img, lbl, label_ready=ready)
        ret["loss"].backward()
        mgr.optimiser.step()
        mgr.scheduler.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    log = Log(int(a.min_mb * 1e6))
    with log:
        step()
    torch.cuda.synchronize()
    lines = [f"# config {a.config}: aten operators on tensors >= {a.min_mb} MB in one training step (count, op, shapes[/strides], where)"]
    for (name, desc, where), (n, b) in sorted(log.rows.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{n:4d} x {b / n / 1e6:7.1f} MB  {name:34s} {desc}\n          {where}")
    txt = "\n".join(lines)
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
