"""How long does the main HIP stream sit idle between the last kernel of training step i and the stem convolution of step i + 1?

rocprofv3's trace of a step shows the main queue idle for ~3.8 ms at the start of every PROFILED step (profiles/r05_step_timeline.txt):
the label histogram -> counts to the host -> host sampling plan -> rank-select chain of the contrastive loss (reference
losses/DenseContrastiveLossV2.py:86-125, issued by managers/HRNet_Manager.py forward_step through LossWrapper.prepare) blocks the host
before it launches the model's forward.  Under the profiler the host is slow and never gets ahead of the GPU; without it the host
issues a step in a fraction of the step's GPU time, so prepare() of step i + 1 runs while the GPU still works on step i.  This tool
measures that claim with HIP events and NO profiler: event A behind the last launch of step i (main stream), event C right in front
of the stem convolution's launch of step i + 1 (forward pre-hook, main stream).  elapsed(A, C) is the time the main stream had
nothing to do between the steps.

    python tools/step_boundary.py [--steps 20] [bench.py arguments]      ->  one line per step + median / max, as text"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    if "--no-cpu-baseline" not in sys.argv:
        sys.argv += ["--no-cpu-baseline"]
    args = bench.parse()
    dev = torch.device("cuda:0")
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

    first = next(m for m in mgr.model.modules() if isinstance(m, torch.nn.Conv2d))     # the stem's convolution on the image
    starts, ends, host = [], [], []

    def pre_hook(_m, _inp):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        starts.append(e)

    first.register_forward_pre_hook(pre_hook)

    def step():
        mgr.optimiser.zero_grad(set_to_none=True)
        ret = mgr.forward_step(img, lbl, label_ready=ready)
        ret["loss"].backward()
        mgr.optimiser.step()
        mgr.scheduler.step()
        if not args.no_metrics:
            mgr.step_metrics(1, ret, lbl, 0.0)
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        ends.append(e)

    import time
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    del starts[:], ends[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        h0 = time.perf_counter()
        step()
        host.append((time.perf_counter() - h0) * 1e3)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3 / args.steps
    gaps = [ends[i].elapsed_time(starts[i + 1]) for i in range(args.steps - 1)]
    steps = [starts[i].elapsed_time(ends[i]) for i in range(args.steps)]
    print(f"# {bench.workload_name(args, 'step')}")
    print(f"# {args.steps} steps, no profiler: wall {wall:.3f} ms / step; host issue time per step median {statistics.median(host):.1f} ms")
    print("# step  main-stream idle between the end of step i and the stem convolution of step i + 1 (ms) | stem .. end of step i (ms)")
    for i, g in enumerate(gaps):
        print(f"{i:4d}  {g:8.3f}  {steps[i]:8.3f}")
    print(f"# idle gap: median {statistics.median(gaps):.3f} ms, mean {statistics.fmean(gaps):.3f}, max {max(gaps):.3f}, min {min(gaps):.3f}")


if __name__ == "__main__":
    main()
