"""fp32 Linear forward / data gradient of the Swin-T shapes (batch 16, 512^2) under the two BLAS back ends torch can
route to (hipBLASLt vs rocBLAS), and with TF32-like modes off: which one should the port ask for?"""
import torch
from torch.nn import functional as F
dev = torch.device("cuda:0")
def timeit(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
shapes = [(262144, 96, 288), (262144, 96, 96), (262144, 96, 384), (262144, 384, 96), (65536, 192, 576), (65536, 192, 192),
          (65536, 192, 768), (65536, 768, 192), (16384, 384, 1152), (16384, 384, 384), (16384, 384, 1536),
          (16384, 1536, 384), (4096, 768, 2304), (4096, 768, 768), (4096, 768, 3072), (4096, 3072, 768)]
for lib in ("hipblaslt", "hipblas"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print(lib, "unavailable:", e); continue
    tot_f = tot_d = 0.0
    for (M, K, N) in shapes:
        x = torch.randn(M, K, device=dev); dy = torch.randn(M, N, device=dev); w = torch.randn(N, K, device=dev)
        b = torch.randn(N, device=dev)
        tf = timeit(lambda: F.linear(x, w, b)); td = timeit(lambda: dy.mm(w))
        tot_f += tf; tot_d += td
        print(f"{lib:10s} M={M} K={K} N={N}: fwd {tf:.0f} us dgrad {td:.0f} us", flush=True)
    print(f"{lib}: sum fwd {tot_f:.0f} us, dgrad {tot_d:.0f} us", flush=True)
