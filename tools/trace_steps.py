"""Per-step wall times of the loss + allocator statistics (run on the GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd  # noqa
from mscs_amd.losses import DenseContrastiveLossV2_ms
from mscs_amd.utils import set_verbosity
set_verbosity(40)
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
n, H, W = 12, 512, 1024
label = torch.randint(0, 20, (n, H, W), generator=gen).to(dev)
feats = [torch.randn(n, 256, H // (4 << s), W // (4 << s), generator=gen).to(dev).requires_grad_(True) for s in range(3)]
mod = DenseContrastiveLossV2_ms({"dataset": "CITYSCAPES", "experiment": 1, "temperature": 0.1, "scales": 3,
                                 "weights": [1.0, 0.7, 0.4], "cross_scale_contrast": True})
def stats():
    s = torch.cuda.memory_stats()
    return s["num_alloc_retries"], s["allocation.all.allocated"], s["segment.all.allocated"], s["reserved_bytes.all.current"] >> 20
for i in range(14):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in feats: f.grad = None
    loss = mod(label, feats)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"step {i:2d} fwd_host {1e3*(t1-t0):6.2f} fwd_gpu_tail {1e3*(t2-t1):6.2f} bwd_host {1e3*(t3-t2):6.2f} bwd_gpu_tail {1e3*(t4-t3):6.2f} total {1e3*(t4-t0):6.2f}  alloc {stats()}")
# no intermediate syncs
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10):
    for f in feats: f.grad = None
    mod(label, feats).backward()
torch.cuda.synchronize(); print("pipelined ms/step", (time.perf_counter() - t0) * 100)
import gc
gc.disable()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10):
    for f in feats: f.grad = None
    mod(label, feats).backward()
torch.cuda.synchronize(); print("pipelined, gc off ms/step", (time.perf_counter() - t0) * 100)
