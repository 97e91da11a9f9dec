"""Micro-benchmark (GPU box): weight gradient of HRNet's cls_head conv (3x3, 720->720, 12x128x256):
MIOpen's default solver vs unfold + rocBLAS batched GEMM."""
import time
import torch
import torch.nn.functional as F

dev = torch.device("cuda:0")
N, C, H, W = 12, 720, 128, 256
x = torch.randn(N, C, H, W, device=dev)
w = torch.randn(C, C, 3, 3, device=dev) * 0.01
gy = torch.randn(N, C, H, W, device=dev)


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


def miopen():
    return torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                               (False, True, False))[1]


def gemm():
    cols = F.unfold(x, 3, padding=1)                       # [N, C*9, HW]
    return torch.bmm(gy.flatten(2), cols.transpose(1, 2)).sum(0).view_as(w)


def gemm_chunked(chunk=4):
    acc = None
    for i in range(0, N, chunk):
        cols = F.unfold(x[i:i + chunk], 3, padding=1)
        part = torch.bmm(gy[i:i + chunk].flatten(2), cols.transpose(1, 2)).sum(0)
        acc = part if acc is None else acc + part
    return acc.view_as(w)


t1, a = timeit(miopen)
t2, b = timeit(gemm)
t3, c = timeit(gemm_chunked)
print(f"miopen wrw {t1:.1f} ms | unfold+bmm {t2:.1f} ms | chunked {t3:.1f} ms | rel err {((a-b).abs().max()/a.abs().max()).item():.2e} {((a-c).abs().max()/a.abs().max()).item():.2e}")
t4, _ = timeit(lambda: F.unfold(x, 3, padding=1))
print(f"unfold alone {t4:.1f} ms; peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GB")
