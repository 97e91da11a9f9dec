"""BN backward per launch: the two-kernel form against the one-kernel form (csrc/dcl_bn_onepass.hip), alone on the GPU."""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import __graft_entry__  # noqa: F401,E402
import mscs_amd.models.fused_bn as fb  # noqa: E402

dev = "cuda"
for shape in ((12, 48, 128, 256), (12, 96, 64, 128), (12, 64, 128, 256), (12, 256, 128, 256), (12, 720, 128, 256)):
    for (res, relu) in ((False, True), (True, True)):
        row = []
        for onepass in (False, True):
            fb.ONEPASS = onepass
            bn = fb.FusedBatchNorm2d(shape[1]).to(dev).train()
            x = torch.randn(shape, device=dev).requires_grad_(True)
            r = torch.randn(shape, device=dev).requires_grad_(True) if res else None
            gy = torch.randn(shape, device=dev)
            y = bn(x, residual=r, relu=relu)
            y.backward(gy, retain_graph=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                y.backward(gy, retain_graph=True)
            e1.record()
            torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f"{shape} residual {res}: two kernels {row[0]:7.1f} us   one kernel {row[1]:7.1f} us", flush=True)
fb.ONEPASS = True
