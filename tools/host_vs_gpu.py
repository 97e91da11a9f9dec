"""Host enqueue time against GPU time of the benchmark step (is the step host-bound?), optional cProfile.
    python tools/host_vs_gpu.py [--config 4] [--profile] [--lead]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import mscs_amd
from mscs_amd.managers import HRNetManager, OCRNetManager
from mscs_amd.utils import set_verbosity
set_verbosity(40)
C4 = "--config" in sys.argv and sys.argv[sys.argv.index("--config") + 1] == "4"
class A:
    batch, height, width, scales, no_cross, channels_last, branch_conv = 12, 512, 1024, 3, False, False, "f16x3"
    materialize_logits, head_conv, config, classes = False, "direct", 2, 20
if C4:
    A.batch, A.height, A.width, A.scales, A.config, A.classes = 16, 512, 512, 4, 4, 151
mgr = (OCRNetManager if C4 else HRNetManager)(bench.step_config(A, 1), autostart=False); mgr.setup(); mgr.model.train()
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
img = torch.randn(A.batch, 3, A.height, A.width, generator=gen).to(dev)
lbl = torch.randint(0, A.classes, (A.batch, A.height, A.width), generator=gen).to(dev)
torch.cuda.synchronize()
ready = torch.cuda.Event(); ready.record()
def step():
    mgr.optimiser.zero_grad(set_to_none=True)
    t0 = time.perf_counter()
    ret = mgr.forward_step(img, lbl, label_ready=ready)
    t1 = time.perf_counter()
    ret["loss"].backward()
    t2 = time.perf_counter()
    mgr.optimiser.step(); mgr.scheduler.step()
    mgr.step_metrics(1, ret, lbl, 0.0)
    return t1 - t0, t2 - t1, time.perf_counter() - t2
for _ in range(4): step()
torch.cuda.synchronize()
T0 = time.perf_counter(); acc = [0, 0, 0]
for _ in range(10):
    a, b, c = step(); acc[0] += a; acc[1] += b; acc[2] += c
host = time.perf_counter() - T0
torch.cuda.synchronize()
tot = time.perf_counter() - T0
print(f"host enqueue {host*100:.1f} ms/step (fwd {acc[0]*100:.1f} bwd {acc[1]*100:.1f} opt {acc[2]*100:.1f}), wall incl. GPU drain {tot*100:.1f} ms/step")
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        step()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
if "--lead" in sys.argv:
    # GPU time of a step when the host is a whole step AHEAD: park the main stream behind a spin kernel while the host
    # enqueues everything, then time from the end of the spin to the end of the step.  The gap to the ordinary step time
    # is what the host's enqueue rate costs (queue depth for the per-branch streams).
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(100_000_000); e1.record(); torch.cuda.synchronize()
    per_cycle = e0.elapsed_time(e1) / 1e8
    cycles = int(160.0 / per_cycle)
    res = []
    for _ in range(5):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(cycles)
        a.record()
        t0 = time.perf_counter()
        step()
        th = time.perf_counter() - t0
        b.record()
        torch.cuda.synchronize()
        res.append((a.elapsed_time(b), th * 1e3))
    print("GPU ms with the host a step ahead (spin %.0f ms):" % (cycles * per_cycle),
          ", ".join("%.1f (host %.0f)" % r for r in res))
