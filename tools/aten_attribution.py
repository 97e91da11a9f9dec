"""Which Python lines launch the ATen (library element-wise / copy / reduce) kernels of a training step?

    python tools/aten_attribution.py --config 5 [--out gpurun_out/aten_attr_config5.txt]

One steady-state step of bench.py's workload under torch.profiler (with_stack, record_shapes); prints the aten:: operators
with device time, grouped by (operator, input shapes, innermost repository frames), sorted by device time.  The kernels of
this package's own library are launched through ctypes and do not appear as operators: what is listed is exactly what is
still left to the framework's element-wise kernels."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=5)
    ap.add_argument("--out", default=None)
    ap.add_argument("--top", type=int, default=60)
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
        mgr.step_metrics(1, ret, lbl, 0.0)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    rows = {}
    for e in prof.events():
        dt = getattr(e, "self_device_time_total", 0) or 0
        if dt <= 0 or not e.name.startswith("aten::"):
            continue
        frames = [f for f in (e.stack or []) if "/repo/" in f or "mscs_amd" in f or "eccv2022" in f][:3]
        frames = [f.replace(ROOT + "/", "") for f in frames]
        key = (e.name, str(e.input_shapes)[:120], " <- ".join(frames))
        r = rows.setdefault(key, [0, 0.0])
        r[0] += 1
        r[1] += dt
    out = []
    tot = sum(r[1] for r in rows.values())
    out.append(f"# config {a.config}: {sum(r[0] for r in rows.values())} aten operators with device time, {tot / 1e3:.2f} ms in one step")
    for (name, shapes, frames), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:a.top]:
        out.append(f"{t / 1e3:8.3f} ms {n:4d} x {name:28s} {shapes}\n             {frames}")
    txt = "\n".join(out)
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
