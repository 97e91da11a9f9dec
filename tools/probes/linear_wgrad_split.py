"""How far is the library's fp32 weight-gradient GEMM of a token-major Linear (dW = dY^T X, reduction over 10^5 tokens, a
tiny output) from the HBM bound, and does cutting the token axis into batched slabs help?"""
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
for (M, K, N) in [(262144, 96, 288), (262144, 96, 96), (262144, 96, 384), (262144, 384, 96), (65536, 192, 576),
                  (65536, 192, 768), (65536, 768, 192), (16384, 384, 1152), (16384, 384, 1536), (16384, 1536, 384),
                  (4096, 768, 2304), (4096, 768, 3072)]:
    x = torch.randn(M, K, device=dev); dy = torch.randn(M, N, device=dev); w = torch.randn(N, K, device=dev)
    hbm = (M * K + M * N) * 4 / 4.5e12 * 1e6
    t0 = timeit(lambda: dy.t().mm(x))
    line = f"M={M} K={K} N={N}: wgrad {t0:.0f} us (HBM bound {hbm:.0f})"
    for s in (16, 64, 256):
        if M // s >= 256:
            t = timeit(lambda: torch.bmm(dy.view(s, M // s, N).transpose(1, 2), x.view(s, M // s, K)).sum(0))
            line += f" | {s} slabs {t:.0f}"
    tf = timeit(lambda: F.linear(x, w)); td = timeit(lambda: dy.mm(w))
    fl = 2.0 * M * K * N
    line += f" || fwd {tf:.0f} us ({fl / tf / 1e6:.0f} TF) dgrad {td:.0f} us ({fl / td / 1e6:.0f} TF), wgrad {fl / t0 / 1e6:.0f} TF"
    print(line, flush=True)
