"""Accuracy of both window-attention forward kernels against float64 on windows without padding / shift."""
import os
import sys

import torch

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import __graft_entry__  # noqa: F401,E402
from mscs_amd import _lib  # noqa: E402
from mscs_amd.models import ops  # noqa: E402

L = _lib.lib()
dev = "cuda"
for (B, H, W, heads, std) in ((2, 14, 14, 3, 1.0), (16, 70, 70, 6, 1.0), (16, 70, 70, 6, 3.0), (16, 161, 161, 6, 1.0)):
    C = 32 * heads
    torch.manual_seed(1)
    qkv = torch.randn(B, H * W, 3 * C, device=dev) * std
    qb = torch.zeros(3 * C, device=dev)
    bias = torch.randn(heads, 49, 49, device=dev) * 0.5
    x = qkv.double().view(B, H // 7, 7, W // 7, 7, 3, heads, 32).permute(5, 0, 1, 3, 6, 2, 4, 7).reshape(3, -1, heads, 49, 32)
    q, k, v = x[0] * 32 ** -0.5, x[1], x[2]
    att = (q @ k.transpose(-1, -2) + bias.double()).softmax(-1) @ v                   # [B nW, heads, 49, 32]
    ref = att.view(B, H // 7, W // 7, heads, 7, 7, 32).permute(0, 1, 4, 2, 5, 3, 6).reshape(B, H * W, C)
    for mask in (0, 1):
        L.dcl_winattn_set_mfma(mask)
        out = ops.window_attention(qkv, qb, bias, H, W, heads, 0, 32 ** -0.5)
        err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
        print(f"B {B} {H}x{W} heads {heads} std {std}: mask {mask} err vs fp64 {err:.2e}")
L.dcl_winattn_set_mfma(3)
