"""Operator-level replacements inside the models where the library's default kernel is far from the hardware roofline on MI355X
(measured, see profiles/).  Since round 5 one module per operator family; this module re-exports all of them, so
``from .ops import X`` / ``ops.X`` keep working:

    ops_resize   bilinear up-sampling (+ add, + ReLU), concatenation without copies, one-kernel gradient sums (fan_out)
    ops_conv     direct f16x3 3x3 / 1x1 convolutions (forward, data gradient, weight gradient), DirectConv2d, weight packing
    ops_conv1x1  1x1 convolutions as batched split-f16 GEMMs (NCHW, token-major input, Dropout2d folded in, pixel-major output)
    ops_head     the head convolution over up-sampled maps without the up-sampled maps (tap products + tap gather)
    ops_linear   dcl_gemm_f16x3 and the token-major Linears (fused Mlp, residual epilogues, GELU)
    ops_swin     LayerNorm, window attention
    ops_logits   fused up-sampling + cross-entropy (UpsampledLogits)

Module switches (A/B runs, tests) live in the module that reads them: ``ops_resize.HIP_UPSAMPLE``,
``ops_linear.FUSED_MLP`` ... -- assigning to the copy in THIS namespace changes nothing."""
from . import ops_common, ops_resize, ops_linear, ops_conv, ops_conv1x1, ops_head, ops_swin, ops_logits

for _m in (ops_common, ops_resize, ops_linear, ops_conv, ops_conv1x1, ops_head, ops_swin, ops_logits):
    globals().update({_k: _v for _k, _v in vars(_m).items() if not _k.startswith("__")})
del _m
