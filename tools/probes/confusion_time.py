"""Time of the metrics tail's histogram kernel at the benchmark size."""
import sys, torch
sys.path.insert(0, "/root/repo")
import mscs_amd
from mscs_amd import _lib
dev = torch.device("cuda:0")
total = 12 * 512 * 1024
for name, gen_t in (("iid", lambda: torch.randint(0, 20, (total,), device=dev)),
                    ("blocky", lambda: torch.randint(0, 20, (12, 16, 32), device=dev).repeat_interleave(32, 1).repeat_interleave(32, 2).reshape(-1))):
    t = gen_t()
    p = t.clamp(max=18).to(torch.uint8)
    cm = torch.zeros((19, 20), dtype=torch.int32, device=dev)
    oob = torch.zeros(1, dtype=torch.int32, device=dev)
    f = lambda: _lib.lib().dcl_confusion_matrix_pred(_lib.ptr(p), total, _lib.ptr(t), 8, 19, 20, _lib.ptr(cm), _lib.ptr(oob), _lib.stream_ptr(dev))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    print(name, "us per launch", round(e0.elapsed_time(e1) * 50, 1))

total = 16 * 640 * 640
t = torch.randint(0, 151, (total,), device=dev)
p = torch.randint(0, 150, (total,), device=dev).to(torch.uint8)
cm = torch.zeros((150, 151), dtype=torch.int32, device=dev)
oob = torch.zeros(1, dtype=torch.int32, device=dev)
f = lambda: _lib.lib().dcl_confusion_matrix_pred(_lib.ptr(p), total, _lib.ptr(t), 8, 150, 151, _lib.ptr(cm), _lib.ptr(oob), _lib.stream_ptr(dev))
for _ in range(3):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    f()
e1.record(); torch.cuda.synchronize()
print("ADE20K 150 x 151, 16 x 640 x 640 iid: us per launch", round(e0.elapsed_time(e1) * 50, 1))
