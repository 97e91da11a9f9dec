"""GPU microbench: fp32-equivalent GEMM as three fp16 GEMMs with fp32 output (split hi/lo operands)
versus the plain fp32 GEMM, at the shape of HRNet's head conv as an im2col GEMM."""
import time
import torch

dev = torch.device("cuda:0")
M, K, N = 12 * 128 * 256 // 4, 6480, 720          # one 3-image chunk of the unfolded input
a = torch.randn(M, K, device=dev)
b = torch.randn(K, N, device=dev) * 0.02


def split(x):
    s = 2.0 ** (torch.floor(torch.log2(torch.tensor(30000.0, device=dev) / x.abs().max())))
    xs = x * s
    hi = xs.half()
    lo = (xs - hi.float()).half()
    return hi, lo, s


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


t32, ref = timeit(lambda: a @ b)
flops = 2.0 * M * K * N
print(f"fp32 GEMM {t32:.2f} ms = {flops / t32 / 1e9:.1f} TFLOP/s")
ah, al, sa = split(a)
bh, bl, sb = split(b)
try:
    def f3():
        o = torch.mm(ah, bh, out_dtype=torch.float32)
        o += torch.mm(ah, bl, out_dtype=torch.float32)
        o += torch.mm(al, bh, out_dtype=torch.float32)
        return o / (sa * sb)
    t3, out = timeit(f3)
    err = (out - ref).abs().max().item() / ref.abs().max().item()
    ref64 = (a[:2048].double() @ b.double())
    e32 = (ref[:2048].double() - ref64).abs().max().item() / ref64.abs().max().item()
    e3 = (out[:2048].double() - ref64).abs().max().item() / ref64.abs().max().item()
    print(f"f16x3 (3 x mm out_dtype=f32) {t3:.2f} ms = {flops / t3 / 1e9:.1f} TFLOP/s-equivalent; rel err vs fp32 {err:.2e}; "
          f"vs fp64: fp32 {e32:.2e}, f16x3 {e3:.2e}")
    t1, _ = timeit(lambda: torch.mm(ah, bh, out_dtype=torch.float32))
    print(f"single fp16 mm with f32 out: {t1:.2f} ms = {flops / t1 / 1e9:.1f} TFLOP/s")
except Exception as e:  # noqa: BLE001
    print("out_dtype path failed:", type(e).__name__, str(e)[:200])
t16, _ = timeit(lambda: ah @ bh)
print(f"plain fp16 mm (fp16 out): {t16:.2f} ms = {flops / t16 / 1e9:.1f} TFLOP/s")
tsplit, _ = timeit(lambda: split(a))
print(f"split of A ({M}x{K}): {tsplit:.2f} ms")
