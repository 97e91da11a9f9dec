#!/usr/bin/env python3
"""Time the InfoNCE backward sweep (dcl_infonce_bwd, f16x3) at the benchmark shape for several column-split counts:
separates the per-workgroup fixed cost (A-panel load, slab store) from the per-chunk cost.
    python tools/sweep_scan.py [nsplit ...]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import mscs_amd  # noqa: E402,F401
from mscs_amd import _lib  # noqa: E402
from mscs_amd.losses import DenseContrastiveLossV2_ms  # noqa: E402
from mscs_amd.utils import set_verbosity  # noqa: E402

set_verbosity(40)
dev = torch.device("cuda:0")
args = bench.parse.__globals__["argparse"].Namespace(batch=12, height=512, width=1024, scales=1)
mod = DenseContrastiveLossV2_ms(bench.loss_config(1, False))
label, feats = bench.synth_loss_inputs(args, dev, 0)
torch.manual_seed(0)
mod(label, feats).backward()
L = _lib.lib()
st = mod.last_state
t = st.terms[0]
A = st.scales[0]
N, Npad = A.plan.N, A.bank.shape[0]
stat = torch.empty((Npad + 1, 4), device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = _lib.ptr
_lib.check(L.dcl_infonce_prep_stats(p(t.Z), p(t.W), p(t.rng_lo), p(t.rng_hi), None, N, A.plan.V, 1, 1.0, 1.0 / t.tau,
                                    None, p(stat), stream), "prep")
for ns in [int(x) for x in sys.argv[1:]] or [3, 6, 10, 13, 20, 26]:
    dpart = torch.empty((ns, Npad, 256), device=dev)

    def launch():
        _lib.check(L.dcl_infonce_bwd(p(A.bank), N, A.plan.V, p(A.bank), N, p(t.rng_lo), p(t.rng_hi), 1.0 / t.tau, 1, 1, 1,
                                     p(stat), p(stat), ns, p(dpart), p(A.bank_h), p(A.bank_h), stream), "bwd")
    ms = bench._time_launches(launch, 20)
    wgs = Npad // 128 * ns
    print(f"nsplit {ns:2d}: {wgs:4d} workgroups ({wgs / 256:.2f} rounds), {ms * 1e3:7.1f} us, "
          f"{4.0 * N * N * 256 / ms / 1e9:6.1f} TFLOP/s, chunks per workgroup {(N + 31) // 32 / ns:.1f}", flush=True)
    del dpart

# fixed-cost probe: the same grid (77 row blocks x 13 splits) with ONE chunk of columns per workgroup
ns, n2 = 13, 13 * 32
dpart = torch.empty((ns, Npad, 256), device=dev)
lo0 = torch.zeros_like(t.rng_lo)


def launch1():
    _lib.check(L.dcl_infonce_bwd(p(A.bank), N, A.plan.V, p(A.bank), n2, p(lo0), p(lo0), 1.0 / t.tau, 0, 1, 1,
                                 p(stat), p(stat), ns, p(dpart), p(A.bank_h), p(A.bank_h), stream), "bwd")


ms = bench._time_launches(launch1, 20)
print(f"fixed-cost probe (1 chunk per workgroup, 1001 workgroups): {ms * 1e3:.1f} us -> {ms * 1e3 / 3.91:.1f} us per round")
