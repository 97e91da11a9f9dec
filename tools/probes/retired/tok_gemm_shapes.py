#!/usr/bin/env python3
"""f16x3 token GEMM (csrc/dcl_tokgemm.hip) against the library's fp32 GEMM on the Linear shapes of Swin-T at batch 16,
512 x 512: error against fp64 (a 4096-row sample) and time per call, forward and data gradient.
    python tools/tok_gemm_shapes.py"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mscs_amd  # noqa: E402,F401
from mscs_amd.models import ops  # noqa: E402
from mscs_amd.models.amax import amax_of  # noqa: E402
from bench_conv3x3 import timeit  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
shapes = [(262144, 96, 288), (262144, 96, 96), (262144, 96, 384), (262144, 384, 96), (65536, 192, 576), (65536, 192, 192),
          (65536, 192, 768), (65536, 768, 192), (16384, 384, 1152), (16384, 384, 384), (16384, 384, 1536),
          (16384, 1536, 384), (4096, 768, 2304), (4096, 768, 768), (4096, 768, 3072), (4096, 3072, 768), (1000, 96, 32)]
tot = [0.0, 0.0, 0.0, 0.0]
for (M, K, N) in shapes:
    x = torch.randn(M, K, device=dev) * 1.5
    w = torch.randn(N, K, device=dev) * (1.0 / K) ** 0.5
    b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev) * 1e-3
    sw = amax_of(w)
    wp = ops.conv3x3_pack(w.view(N, K, 1, 1), sw, False)
    wpt = ops.conv3x3_pack(w.view(N, K, 1, 1), sw, True)
    sx, sd = amax_of(x), amax_of(dy)
    y = ops.tok_gemm(x, wp, N, sx, sw, b)
    dx = ops.tok_gemm(dy, wpt, K, sd, sw)
    r = slice(0, min(M, 4096))
    y64 = F.linear(x[r].double(), w.double(), b.double())
    dx64 = dy[r].double().mm(w.double())
    e = lambda a, ref: ((a.double() - ref).abs().max() / ref.abs().max()).item()
    ey, el = e(y[r], y64), e(F.linear(x[r], w, b), y64)
    ed, edl = e(dx[r], dx64), e(dy[r].mm(w), dx64)
    from mscs_amd import _lib
    _lib.lib().dcl_tok_gemm_set_rows(1)
    tf1 = timeit(lambda: ops.tok_gemm(x, wp, N, sx, sw, b), 10) * 1e3
    _lib.lib().dcl_tok_gemm_set_rows(2)
    tf2 = timeit(lambda: ops.tok_gemm(x, wp, N, sx, sw, b), 10) * 1e3
    _lib.lib().dcl_tok_gemm_set_rows(0)
    tf = timeit(lambda: ops.tok_gemm(x, wp, N, sx, sw, b), 10) * 1e3
    tl = timeit(lambda: F.linear(x, w, b), 10) * 1e3
    td = timeit(lambda: ops.tok_gemm(dy, wpt, K, sd, sw), 10) * 1e3
    tdl = timeit(lambda: dy.mm(w), 10) * 1e3
    fl = 2.0 * M * K * N
    tot = [tot[0] + tf, tot[1] + tl, tot[2] + td, tot[3] + tdl]
    print(f"M={M:6d} K={K:4d} N={N:4d}: fwd {tf:6.1f} us ({fl / tf / 1e6:4.0f} TF; P=1 {tf1:6.1f}, P=2 {tf2:6.1f}) library {tl:6.1f} | dgrad {td:6.1f} library {tdl:6.1f} "
          f"| err y {ey:.1e} (lib {el:.1e}) dx {ed:.1e} (lib {edl:.1e})", flush=True)
print("sums: fwd %.0f vs %.0f us, dgrad %.0f vs %.0f us" % tuple(tot))
