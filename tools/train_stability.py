"""Short training run of the benchmark configuration through HRNetManager on one resident synthetic batch: loss per
step (it must fall: the model memorises the batch), reserved memory per step (it must stay flat), step time.
    python tools/train_stability.py [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (reuses the benchmark's manager configuration)
import mscs_amd  # noqa: E402,F401
from mscs_amd.managers import HRNetManager  # noqa: E402
from mscs_amd.utils import set_verbosity  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
set_verbosity(40)
_argv = sys.argv
sys.argv = [sys.argv[0], "--batch", "4"]       # the benchmark's defaults at a smaller batch so that the run finishes quickly
A = bench.parse()
sys.argv = _argv
cfg = bench.step_config(A, 1)
cfg["train"]["learning_rate"] = 0.02
mgr = HRNetManager(cfg, autostart=False)
mgr.setup()
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
img = torch.randn(A.batch, 3, A.height, A.width, generator=gen).to(dev)
lbl = (torch.rand(A.batch, 1, 1, generator=gen) * 19).long().expand(A.batch, A.height, A.width).contiguous()
lbl = (lbl + (torch.arange(A.width) // 128).view(1, 1, -1)) % 19          # vertical stripes of classes
lbl = lbl.to(dev)
mgr.model.train()
for it in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mgr.optimiser.zero_grad(set_to_none=True)
    ret = mgr.forward_step(img, lbl)
    ret["loss"].backward()
    mgr.optimiser.step()
    torch.cuda.synchronize()
    if it % 4 == 0 or it == steps - 1:
        print(f"step {it:3d} loss {ret['loss'].item():9.4f} ms {1e3 * (time.perf_counter() - t0):7.1f} "
              f"reserved {torch.cuda.memory_reserved() >> 20} MiB finite {torch.isfinite(ret['loss']).item()}", flush=True)
