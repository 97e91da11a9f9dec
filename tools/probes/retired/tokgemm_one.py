import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd
from mscs_amd import _lib
from mscs_amd.models import ops
from mscs_amd.models.amax import amax_of
dev = torch.device("cuda:0")
M, K, N = 16384, 384, 1152
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev)
sw, sx = amax_of(w), amax_of(x)
wp = ops.conv3x3_pack(w.view(N, K, 1, 1), sw, False)
for mode in (1, 1 + 16 * 8, 1 + 16 * 4):       # both operands in LDS | weights in LDS | per-wave operands
    _lib.lib().dcl_tok_gemm_set_rows(mode)
    for _ in range(5): ops.tok_gemm(x, wp, N, sx, sw)
torch.cuda.synchronize()
