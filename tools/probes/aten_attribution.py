"""Which Python lines launch the element-wise ATen kernels of a step?  torch.profiler with stacks over one steady-state step of
bench.py's config (default 4), grouped by (op, input shape, innermost repo frame).   python tools/probes/aten_attribution.py [config]"""
import collections
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "4"
sys.argv = ["bench.py", "--config", cfg, "--steps", "1", "--warmup", "2", "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)

import mscs_amd  # noqa: F401,E402
from mscs_amd.managers import HRNetManager, OCRNetManager  # noqa: E402
from mscs_amd.utils import set_verbosity  # noqa: E402

set_verbosity(40)
mgr = (OCRNetManager if args.config in (4, 5) else HRNetManager)(bench.step_config(args, 1), autostart=False)
mgr.setup()
mgr.model.train()
gen = torch.Generator().manual_seed(0)
img = torch.randn(args.batch, 3, args.height, args.width, generator=gen).to(dev)
lbl = torch.randint(0, args.classes, (args.batch, args.height, args.width), generator=gen).to(dev)
ready = torch.cuda.Event()
ready.record()


def step():
    mgr.optimiser.zero_grad(set_to_none=True)
    ret = mgr.forward_step(img, lbl, label_ready=ready)
    ret["loss"].backward()
    mgr.optimiser.step()
    mgr.scheduler.step()
    mgr.step_metrics(1, ret, lbl, 0.0)


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for ka in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
    t = getattr(ka, "self_device_time_total", None)
    if t is None:
        t = getattr(ka, "self_cuda_time_total", 0)
    if not ka.key.startswith("aten::") or t <= 0:
        continue
    frame = next((f for f in (ka.stack or []) if ("mscs_amd" in f or "eccv2022" in f) and "aten_attribution" not in f),
                 next((f for f in (ka.stack or []) if "bench" in f or "torch/optim" in f or "autograd" in f), "?"))
    rows.append((t, ka.count, ka.key, str(ka.input_shapes)[:64], frame[-100:]))
for t, n, name, shp, frame in sorted(rows, reverse=True)[:45]:
    print(f"{t / 1e3:8.2f} ms {n:4d}x {name:26s} {shp:64s} {frame}")
