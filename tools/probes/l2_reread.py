"""Microbenchmark behind the fused BN question: is the second pass over a channel's planes served by the XCD's L2?
   (here)  hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/src/l2_reread.hip -o tools/probes/_build/l2_reread.so"""
import ctypes
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(HERE, "_build", "l2_reread.so"))
dev = "cuda"
for (channels, per, label) in ((48, 12 * 128 * 256, "48 ch x 12 x 128 x 256"), (96, 12 * 64 * 128, "96 ch x 12 x 64 x 128"),
                               (192, 12 * 32 * 64, "192 ch x 12 x 32 x 64")):
    a = torch.randn(channels * per, device=dev)
    b = torch.randn(channels * per, device=dev)
    c = torch.empty_like(a)
    T = 32
    sums = torch.zeros(channels * T, device=dev)
    ctr = torch.zeros(channels, dtype=torch.int32, device=dev)
    L.l2_launch.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    for mode in (0, 3, 1, 2):
        def launch():
            ctr.zero_()
            rc = L.l2_launch(a.data_ptr(), b.data_ptr(), c.data_ptr(), sums.data_ptr(), ctr.data_ptr(), channels, per, mode,
                             torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        try:
            launch()
        except Exception as e:  # noqa: BLE001
            print("launch failed", e)
            break
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            launch()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        mb = a.numel() * 4 / 1e6
        print(f"{label}: mode {mode}: {ms * 1e3:7.1f} us   (one tensor = {mb:.0f} MB)")
