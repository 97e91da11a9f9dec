#!/usr/bin/env python3
"""One forward Linear product on dcl_gemm_f16x3, M K N from the command line, 10 launches (for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mscs_amd  # noqa
from mscs_amd.models import ops
m, k, n = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda:0")
x = torch.randn(m, k, device=dev); w = torch.randn(n, k, device=dev) * k ** -0.5; b = torch.randn(n, device=dev)
for _ in range(10):
    y = ops.linear_f16x3(x, w, b)
torch.cuda.synchronize()
print("ok", tuple(y.shape))
