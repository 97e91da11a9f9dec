"""Is a whole training step (HRNet-W48 + LossWrapper(CE + DenseContrastiveLossV2_ms), manager path, all streams) reproducible
BITWISE from run to run?    python tools/probes/step_repro.py [steps=3] [runs=4] [H=128 W=256 batch=2] [config=2|4|5] [flags=--plain-config,...]

Builds the manager `runs` times from the same seed, takes `steps` optimizer steps on the same resident batch, and compares the
losses of every step and every parameter / buffer after the last one with the first run."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
kv = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
sys.argv = [sys.argv[0], "--height", kv.get("H", "128"), "--width", kv.get("W", "256"), "--batch", kv.get("batch", "2"),
            "--config", kv.get("config", "2"), "--labels", kv.get("labels", "iid")] + [f for f in kv.get("flags", "").split(",") if f]
import bench  # noqa: E402
import mscs_amd  # noqa: F401,E402
from mscs_amd.managers import HRNetManager, OCRNetManager  # noqa: E402
from mscs_amd.utils import set_verbosity  # noqa: E402

set_verbosity(40)
args = bench.parse()
dev = torch.device("cuda:0")
steps, runs = int(kv.get("steps", 3)), int(kv.get("runs", 4))
first = None
ndiff = 0
for r in range(runs):
    torch.manual_seed(0)
    mgr = (OCRNetManager if args.config in (4, 5) else HRNetManager)(bench.step_config(args, 1), autostart=False)
    mgr.setup()
    mgr.model.train()
    gen = torch.Generator().manual_seed(0)
    img = torch.randn(args.batch, 3, args.height, args.width, generator=gen).to(dev)
    lbl = bench.synth_labels(args, args.batch, args.height, args.width, gen).to(dev)
    losses = []
    init = {k: v.detach().clone() for k, v in mgr.model.state_dict().items()}
    if first is not None:
        nb = [k for k in init if not torch.equal(init[k], first_init[k])]
        print(f"run {r}: {len(nb)} initial tensors differ", nb[:4])
    else:
        first_init = init
    for i in range(steps):
        mgr.optimiser.zero_grad(set_to_none=True)
        ret = mgr.forward_step(img, lbl)
        ret["loss"].backward()
        mgr.optimiser.step()
        mgr.scheduler.step()
        losses.append(ret["loss"].detach().clone())
    torch.cuda.synchronize()
    state = {k: v.detach().clone() for k, v in mgr.model.state_dict().items()}
    state.update({f"loss{i}": l for i, l in enumerate(losses)})
    if first is None:
        first = state
        print("losses", [f"{l.item():.6f}" for l in losses])
    else:
        bad = [(k, ((state[k].double() - first[k].double()).abs().max() / (first[k].double().abs().max() + 1e-30)).item())
               for k in first if not torch.equal(state[k], first[k])]
        ndiff += 1 if bad else 0
        print("losses", [f"{l.item():.6f}" for l in losses])
        print(f"run {r}: {len(bad)} of {len(first)} tensors differ", [(k, f"{e:.0e}") for k, e in bad[:6]])
    del mgr
print(f"SUMMARY: {ndiff} of {runs - 1} runs differ from run 0")
