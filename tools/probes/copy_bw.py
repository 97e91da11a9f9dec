"""What does a plain device copy reach on this box at the BN plane sizes (the yardstick for the fused BN kernels)?"""
import torch
dev = torch.device("cuda:0")
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for mb in (9.4, 18.9, 37.7, 75.5, 402.0, 1132.0):
    n = int(mb * 1e6 / 4)
    x = torch.randn(n, device=dev); y = torch.empty_like(x); z = torch.randn(n, device=dev)
    t = timeit(lambda: y.copy_(x))
    t2 = timeit(lambda: torch.add(x, z, out=y))
    t3 = timeit(lambda: x.sum())
    print(f"{mb:7.1f} MB: copy {t*1e3:6.1f} us ({2*mb/t/1e3:4.2f} TB/s)  add {t2*1e3:6.1f} us ({3*mb/t2/1e3:4.2f} TB/s)  sum {t3*1e3:6.1f} us ({mb/t3/1e3:4.2f} TB/s)")
