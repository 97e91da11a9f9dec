"""Which path bounds k_tok_gemm?  Timing with the x loads / the weight loads pinned to chunk 0 (results wrong)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd
from mscs_amd import _lib
from mscs_amd.models import ops
from mscs_amd.models.amax import amax_of
from bench_conv3x3 import timeit
dev = torch.device("cuda:0")
for (M, K, N) in [(16384, 384, 1152), (16384, 1536, 384), (4096, 768, 3072), (65536, 192, 768), (65536, 768, 192), (262144, 96, 288), (262144, 384, 96)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
    sw, sx = amax_of(w), amax_of(x)
    wp = ops.conv3x3_pack(w.view(N, K, 1, 1), sw, False)
    line = f"M={M} K={K} N={N}:"
    for p in (1, 2):
        for dbg, name in ((0, "both in LDS"), (8, "weights in LDS"), (4, "per-wave")):
            _lib.lib().dcl_tok_gemm_set_rows(p + 16 * dbg)
            t = timeit(lambda: ops.tok_gemm(x, wp, N, sx, sw), 10) * 1e3
            line += f" P{p} {name} {t:.0f}"
    print(line, flush=True)
_lib.lib().dcl_tok_gemm_set_rows(0)
